"""World-size-2 / 3 / 8 `gloo` tests (CPU) of the multi-GPU protocol of the embedding loop: row partition ->
local forces -> ONE all-reduce of the message [gradient | loss limbs] -> identical apply on every rank.  The HIP session is replaced by a
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
    """EmbedSession stand-in: same forces_msg/apply_msg contract (kmap_hip.h), rows [row0,row0+nrows) only."""

    def __init__(self, p, ld, row0, nrows, lr, msg_np):
        from oracle import oracle as O
        self.O, self.p, self.ld, self.row0, self.nrows, self.lr = O, p, ld.copy(), row0, nrows, lr
        self.n = p.shape[0]
        self.msg = msg_np
        self.losses, self.prev = [], np.inf

    def forces_msg(self, mp):
        from kmap_amd.distributed import loss_to_limbs
        O, n = self.O, self.n
        q = O.cal_ld_prob_mat(self.ld)
        g = O.gradient_loss(self.p, q, self.ld) / 4.0                      # kernel output before the x4
        r = slice(self.row0, self.row0 + self.nrows)
        self.msg[:2 * n].reshape(2, n)[:, r] = g[:, r]                      # own rows only; the rest must still be zero
        eps, one = np.float32(1e-10), np.float32(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            full = -self.p * np.log(q) - (one - self.p) * np.log(one - q)
            ce = np.where(self.p < eps, -np.log(one - q), np.where(self.p > one - eps, -np.log(q), full))
        self.msg[2 * n:] = loss_to_limbs(np.triu(ce, 1)[r].astype(np.float64).sum())

    def apply_msg(self, mp):
        from kmap_amd.distributed import loss_from_limbs
        n = self.n
        g = self.msg[:2 * n].reshape(2, n).copy()
        cur = np.float32(2.0 * loss_from_limbs(self.msg[2 * n:]))
        self.msg[:2 * n] = 0                                               # apply_msg_kernel<CLEAR>: read, then zero
        self.losses.append(cur)
        if abs(self.prev - cur) < 1e-7 * abs(cur):
            return
        self.prev = cur
        self.ld += (-(4.0 * g) * self.lr)
        self.ld = self.O.add_jitter(self.ld, eps=0.1)


def _worker(rank, world, port, n_iter, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from kmap_amd.distributed import MSG_EXTRA, DistEmbedLoop, row_partition
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
        msg_t = torch.zeros(2 * n + MSG_EXTRA, dtype=torch.float32)
        sess = OracleSession(p, ld, row0, nrows, 0.01, msg_t.numpy())
        loop = DistEmbedLoop(sess, msg_t, dist if world > 1 else None)
        loop.step(n_iter)
        assert loop.n_collectives == (n_iter if world > 1 else 0)          # ONE collective per iteration
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


class CyclicOracleSession:
    """Stand-in for EmbedSession(cyclic=(world, rank)): the rank evaluates each unordered pair (i < j) whose row i lies in
    one of its cyclic 256-row blocks and writes partial gradients for BOTH points; the all-reduce then is a true sum."""

    def __init__(self, p, ld, blocks, lr, msg_np):
        from oracle import oracle as O
        self.O, self.p, self.ld, self.lr = O, p, ld.copy(), lr
        self.msg = msg_np
        n = self.n = p.shape[0]
        own = np.zeros(n, bool)
        for r0, nr in blocks:
            own[r0:r0 + nr] = True
        self.mask = own[:, None] & (np.arange(n)[None, :] > np.arange(n)[:, None])
        self.losses, self.prev = [], np.inf

    def forces_msg(self, mp):
        from kmap_amd.distributed import loss_to_limbs
        O, p, n = self.O, self.p, self.n
        q = O.cal_ld_prob_mat(self.ld).astype(np.float64)
        t = np.where(self.mask, (q / (1 - q)) * (p - q), 0.0)
        g = self.msg[:2 * n].reshape(2, n)
        for c in (0, 1):
            d = self.ld[c][:, None].astype(np.float64) - self.ld[c][None, :]
            f = t * d
            g[c] = (f.sum(axis=1) - f.sum(axis=0)).astype(np.float32)   # row side minus column side: ALL entries overwritten
        ce = -(p * np.log(q) + (1 - p) * np.log(1 - q))
        self.msg[2 * n:] = loss_to_limbs(np.where(self.mask, ce, 0.0).sum())

    def apply_msg(self, mp):
        from kmap_amd.distributed import loss_from_limbs
        n = self.n
        cur = np.float32(2.0 * loss_from_limbs(self.msg[2 * n:]))
        self.losses.append(cur)
        self.prev = cur
        self.ld += (-(4.0 * self.msg[:2 * n].reshape(2, n)) * self.lr)     # apply_msg_kernel<no CLEAR>: nothing zeroed


def _cyclic_worker(rank, world, port, n_iter, out_dir, n=600):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from kmap_amd.distributed import MSG_EXTRA, DistEmbedLoop
    from kmap_amd.visualization import cyclic_blocks
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(4)                          # n = 600: three 256-row blocks: 256, 256, 88
        p = rng.random((n, n)) * 0.9 + 0.05
        p = np.triu(p, 1)
        p = (p + p.T).astype(np.float32)
        ld = (rng.standard_normal((2, n)) * 3).astype(np.float32)
        blocks = cyclic_blocks(n, world, rank)
        msg_t = torch.zeros(2 * n + MSG_EXTRA, dtype=torch.float32)
        sess = CyclicOracleSession(p, ld, blocks, 0.0005, msg_t.numpy())   # small steps: no chaotic amplification
        loop = DistEmbedLoop(sess, msg_t, dist if world > 1 else None)
        loop.step(n_iter)
        np.savez(Path(out_dir) / f"cyc{rank}_of{world}_n{n}.npz", ld=sess.ld, losses=np.array(sess.losses, np.float32),
                 blocks=np.array(blocks).reshape(-1, 2))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_cyclic_block_protocol_matches_single_process(tmp_path):
    """the cyclic symmetric layout (FAST from N = 16384 in the product): block dealing, partial gradients for all points, the
    all-reduce as a true sum -- two and three ranks (one of them with a single ragged block) against one process"""
    import torch.multiprocessing as mp
    n_iter = 6
    for world in (1, 2, 3):
        mp.spawn(_cyclic_worker, args=(world, _free_port(), n_iter, str(tmp_path)), nprocs=world, join=True)
    single = np.load(tmp_path / "cyc0_of1_n600.npz")
    assert single["blocks"].tolist() == [[0, 256], [256, 256], [512, 88]]
    two = [np.load(tmp_path / f"cyc{r}_of2_n600.npz") for r in range(2)]
    assert two[0]["blocks"].tolist() == [[0, 256], [512, 88]] and two[1]["blocks"].tolist() == [[256, 256]]
    for world in (2, 3):
        rs = [np.load(tmp_path / f"cyc{r}_of{world}_n600.npz") for r in range(world)]
        for r in rs[1:]:
            np.testing.assert_array_equal(rs[0]["ld"], r["ld"])        # every rank holds the same iterate
            np.testing.assert_array_equal(rs[0]["losses"], r["losses"])
        np.testing.assert_allclose(rs[0]["losses"], single["losses"], rtol=1e-6)
        np.testing.assert_allclose(rs[0]["ld"], single["ld"], rtol=0, atol=1e-4 * np.abs(single["ld"]).max())


@pytest.mark.timeout(900)
def test_eight_rank_protocols_match_single_process(tmp_path):
    """the node the path is built for has EIGHT ranks: contiguous rows (SEQ; 96 rows = 12 per rank) bit for bit against one process
    and against the golden trace; the cyclic symmetric layout with 9 blocks (rank 0 owns the first and the ragged last one, every
    other rank one) against one process; ragged all-gathers in which three of the eight ranks bring nothing"""
    import torch.multiprocessing as mp
    n_iter = 12
    mp.spawn(_worker, args=(8, _free_port(), n_iter, str(tmp_path)), nprocs=8, join=True)
    mp.spawn(_worker, args=(1, _free_port(), n_iter, str(tmp_path)), nprocs=1, join=True)
    rs = [np.load(tmp_path / f"rank{r}_of8.npz") for r in range(8)]
    single = np.load(tmp_path / "rank0_of1.npz")
    assert [list(r["rows"]) for r in rs] == [[12 * r, 12] for r in range(8)]
    for r in rs[1:]:
        np.testing.assert_array_equal(rs[0]["ld"], r["ld"])
        np.testing.assert_array_equal(rs[0]["losses"], r["losses"])
    np.testing.assert_array_equal(rs[0]["ld"], single["ld"])              # x + 0 + ... + 0: the sum of eight messages is exact
    np.testing.assert_allclose(rs[0]["losses"], single["losses"], rtol=1e-6)
    u = np.load(ROOT / "tests" / "golden" / "umap_n96.npz")
    np.testing.assert_allclose(rs[0]["ld"], u["coords"][n_iter - 1], rtol=0, atol=1e-5)
    # cyclic: 2100 rows = eight full 256-row blocks + one of 52
    n, it = 2100, 3
    mp.spawn(_cyclic_worker, args=(8, _free_port(), it, str(tmp_path), n), nprocs=8, join=True)
    mp.spawn(_cyclic_worker, args=(1, _free_port(), it, str(tmp_path), n), nprocs=1, join=True)
    cs = [np.load(tmp_path / f"cyc{r}_of8_n{n}.npz") for r in range(8)]
    one = np.load(tmp_path / f"cyc0_of1_n{n}.npz")
    assert cs[0]["blocks"].tolist() == [[0, 256], [2048, 52]] and all(cs[r]["blocks"].tolist() == [[256 * r, 256]] for r in range(1, 8))
    for c in cs[1:]:
        np.testing.assert_array_equal(cs[0]["ld"], c["ld"])
        np.testing.assert_array_equal(cs[0]["losses"], c["losses"])
    np.testing.assert_allclose(cs[0]["losses"], one["losses"], rtol=1e-6)
    np.testing.assert_allclose(cs[0]["ld"], one["ld"], rtol=0, atol=1e-4 * np.abs(one["ld"]).max())
    # ragged gathers on eight ranks: five reads (ranks 5..7 own none), rank 1 without a hit
    mp.spawn(_gather_worker, args=(8, _free_port(), str(tmp_path), 5), nprocs=8, join=True)
    gs = [np.load(tmp_path / f"gather{r}.npz") for r in range(8)]
    assert [len(g["mine_hits"]) for g in gs] == [1, 1, 1, 1, 1, 0, 0, 0]
    for key, mine in (("hits", "mine_hits"), ("pos", "mine_pos"), ("nb", "mine_nb")):
        want = np.concatenate([g[mine] for g in gs])
        for g in gs:
            np.testing.assert_array_equal(g[key], want)
            assert g[key].dtype == want.dtype and g[key].shape == want.shape


def _gather_worker(rank, world, port, out_dir, n_reads=1003):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from kmap_amd.distributed import all_gather_concat, broadcast_seed, read_partition
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        # scan-hit shaped payloads: per-read counts (length known from the read partition) and ragged position lists
        borders = np.zeros((n_reads, 2), np.int64)
        lens = [read_partition(borders, world, r)[1] for r in range(world)]
        hits = rng.integers(0, 5, size=lens[rank]).astype(np.int32)
        pos = rng.integers(0, 1 << 20, size=int(hits.sum()) if rank != 1 else 0).astype(np.int32)   # rank 1: no hits at all
        nb = rng.integers(0, n_reads, size=(lens[rank], 20)).astype(np.int32)
        got = {"hits": all_gather_concat(dist, hits, lens), "pos": all_gather_concat(dist, pos), "nb": all_gather_concat(dist, nb, lens),
               "u16": all_gather_concat(dist, (hits * 1000).astype(np.uint16)),
               "seed_none": broadcast_seed(dist, None), "seed_given": broadcast_seed(dist, 41),
               "mine_hits": hits, "mine_pos": pos, "mine_nb": nb}
        np.savez(Path(out_dir) / f"gather{rank}.npz", **got)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_all_gather_concat_and_seed_broadcast(tmp_path):
    """the tensor collectives that replaced all_gather_object: ragged arrays (incl. an empty one) concatenate in rank order
    with dtype and trailing shape intact; a "default" seed becomes one shared seed, a given seed is left alone"""
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"gather{r}.npz") for r in range(world)]
    for key, mine in (("hits", "mine_hits"), ("pos", "mine_pos"), ("nb", "mine_nb")):
        want = np.concatenate([r[mine] for r in rs])
        for r in rs:
            np.testing.assert_array_equal(r[key], want)
            assert r[key].dtype == want.dtype and r[key].shape == want.shape
    assert rs[0]["nb"].shape == (1003, 20) and len(rs[1]["mine_pos"]) == 0
    want16 = np.concatenate([(r["mine_hits"] * 1000).astype(np.uint16) for r in rs])
    for r in rs:
        np.testing.assert_array_equal(r["u16"], want16)
        assert r["u16"].dtype == np.uint16
        assert int(r["seed_none"]) == int(rs[0]["seed_none"]) and 0 <= int(r["seed_none"]) < 2 ** 32
        assert int(r["seed_given"]) == 41


def test_loss_limbs_are_exact_and_order_free():
    """the loss partial travels as integer limbs inside the float32 message: float32 sums of the limbs of many ranks are exact in
    any order, so every rank decodes the same total; non-finite / negative / huge partials raise the flag -> NaN"""
    from kmap_amd.distributed import MSG_EXTRA, loss_from_limbs, loss_to_limbs
    rng = np.random.default_rng(0)
    for scale in (1e-6, 1.0, 1e5, 1e11):
        parts = rng.random(64) * scale
        limbs = np.stack([loss_to_limbs(v) for v in parts])
        assert limbs.shape == (64, MSG_EXTRA) and limbs.dtype == np.float32
        fwd = limbs.sum(axis=0, dtype=np.float32)
        acc = np.zeros(MSG_EXTRA, np.float32)
        for row in limbs[rng.permutation(64)]:
            acc = (acc + row).astype(np.float32)            # a different summation order, float32 all the way
        np.testing.assert_array_equal(fwd, acc)
        got, want = loss_from_limbs(fwd), float(np.sum(parts))
        assert abs(got - want) <= 64 * 2.0 ** -48 + 4e-16 * want        # truncation at 2^-48 per rank + the f64 reference sum's own rounding
    one = 123456.789012345678
    assert loss_from_limbs(loss_to_limbs(one)) == one                   # >= 16: the f64 mantissa fits the 48 fractional bits
    assert loss_from_limbs(loss_to_limbs(0.0)) == 0.0
    for bad in (float("nan"), float("inf"), -1.0, 2.0 ** 47):
        t = loss_to_limbs(bad) + loss_to_limbs(3.0)
        assert np.isnan(loss_from_limbs(t))
