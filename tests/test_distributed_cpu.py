"""World-size-2 `gloo` test (CPU) of the multi-GPU protocol of the embedding loop: row partition ->
local forces -> all-reduce(grad, loss) -> identical apply on every rank.  The HIP session is replaced by a
test double that computes with the CPU oracle (tests may use the oracle; the product never does)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


class OracleSession:
    """EmbedSession stand-in: same forces/apply contract, rows [row0,row0+nrows) only."""

    def __init__(self, p, ld, row0, nrows, lr, grad_np, loss_np):
        from oracle import oracle as O
        self.O, self.p, self.ld, self.row0, self.nrows, self.lr = O, p, ld.copy(), row0, nrows, lr
        self.g, self.l = grad_np, loss_np
        self.losses, self.prev = [], np.inf

    def forces(self, gp, lp):
        O = self.O
        q = O.cal_ld_prob_mat(self.ld)
        g = O.gradient_loss(self.p, q, self.ld) / 4.0                      # kernel output before the x4
        r = slice(self.row0, self.row0 + self.nrows)
        self.g[:, r] = g[:, r]
        eps, one = np.float32(1e-10), np.float32(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            full = -self.p * np.log(q) - (one - self.p) * np.log(one - q)
            ce = np.where(self.p < eps, -np.log(one - q), np.where(self.p > one - eps, -np.log(q), full))
        self.l[0] = np.triu(ce, 1)[r].astype(np.float64).sum()

    def apply(self, gp, lp):
        cur = np.float32(2.0 * self.l[0])
        self.losses.append(cur)
        if abs(self.prev - cur) < 1e-7 * abs(cur):
            return
        self.prev = cur
        self.ld += (-(4.0 * self.g) * self.lr)
        self.ld = self.O.add_jitter(self.ld, eps=0.1)


def _worker(rank, world, port, n_iter, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from kmap_amd.distributed import DistEmbedLoop, row_partition
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        u = np.load(ROOT / "tests" / "golden" / "umap_n96.npz")
        k, n = int(u["kmer_len"]), 96
        S = O.knn_smooth(u["D"].astype(np.int64), 20, nb=u["nb"])
        p = O.hd_prob_from_smooth(S, k)
        np.random.seed(int(u["seed"]))
        ld = np.random.randn(2, n).astype("float32")
        for _ in range(10):
            np.random.randn(2, n)
        row0, nrows = row_partition(n, world, rank)
        grad_t = torch.zeros((2, n), dtype=torch.float32)
        loss_t = torch.zeros(1, dtype=torch.float64)
        sess = OracleSession(p, ld, row0, nrows, 0.01, grad_t.numpy(), loss_t.numpy())
        loop = DistEmbedLoop(sess, grad_t, loss_t, dist if world > 1 else None)
        loop.step(n_iter)
        assert loop.n_collectives == (2 * n_iter if world > 1 else 0)
        np.savez(Path(out_dir) / f"rank{rank}_of{world}.npz", ld=sess.ld, losses=np.array(sess.losses, np.float32),
                 rows=np.array([row0, nrows]))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_two_rank_embedding_protocol_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    n_iter = 25
    mp.spawn(_worker, args=(2, _free_port(), n_iter, str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_worker, args=(1, _free_port(), n_iter, str(tmp_path)), nprocs=1, join=True)
    r0, r1 = np.load(tmp_path / "rank0_of2.npz"), np.load(tmp_path / "rank1_of2.npz")
    single = np.load(tmp_path / "rank0_of1.npz")
    assert list(r0["rows"]) == [0, 48] and list(r1["rows"]) == [48, 48]
    np.testing.assert_array_equal(r0["ld"], r1["ld"])                  # every rank holds the same iterate
    np.testing.assert_array_equal(r0["losses"], r1["losses"])
    np.testing.assert_array_equal(r0["ld"], single["ld"])              # sharded == unsharded, bit for bit
    np.testing.assert_allclose(r0["losses"], single["losses"], rtol=1e-6)
    # and both follow the reference trace (golden fixture) within the float-path tolerance
    u = np.load(ROOT / "tests" / "golden" / "umap_n96.npz")
    np.testing.assert_allclose(r0["ld"], u["coords"][n_iter - 1], rtol=0, atol=1e-5)
    np.testing.assert_allclose(r0["losses"], u["losses"][:n_iter], rtol=2e-6)
