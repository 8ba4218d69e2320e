"""GPU test of the row-sharded path with the real HIP sessions: two ranks share the one GPU of the test box
(gloo transports the CUDA tensors), each owns half of the rows.  Sharded result == unsharded result."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
N, K, ITERS, SEED = 1501, 8, 12, 5


def _inputs():
    rng = np.random.default_rng(31)
    kh = rng.integers(0, 4 ** K, size=N, dtype=np.uint64)
    lab = np.sort(rng.integers(0, 3, size=N)).astype(np.int64)
    if os.environ.get("KMAP_TEST_REPEATS"):      # counts of 1 .. 6: runs of equal rows (the row map of the SEQ sessions)
        return kh, rng.integers(1, 7, size=N).astype(np.int64), lab, ["ACGTACGT", "ACGTAC"]
    return kh, np.ones(N, np.int64), lab, ["ACGTACGT", "ACGTAC"]


def _worker(rank, world, port, mode, out_dir, exchange="rccl"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KMAP_DIST_EXCHANGE=exchange)
    import torch
    import torch.distributed as dist
    from kmap_amd.distributed import kmap_from_kmers_distributed
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kh, cnts, lab, conseqs = _inputs()
        tr = {}
        best, _ = kmap_from_kmers_distributed(kh, cnts, lab, conseqs, K, n_max_iter=ITERS, random_seed=SEED, mode=mode, trace=tr)
        assert tr["exchange"] == ("direct" if exchange == "direct" else "all_reduce")
        np.savez(Path(out_dir) / f"m{mode}_rank{rank}.npz", best=best, last=tr["last_coords"], losses=tr["losses"],
                 d_rows=tr["hbm"]["d_rows"])
    finally:
        dist.destroy_process_group()


def _timeout_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KMAP_DIST_EXCHANGE="direct", KMAP_PEER_TIMEOUT_MS="2000")
    import time
    import torch
    import torch.distributed as dist
    from kmap_amd.distributed import kmap_from_kmers_distributed
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kh, cnts, lab, conseqs = _inputs()
        t0, msg = time.perf_counter(), ""
        try:      # rank 1 runs no iteration, i.e. never pushes: rank 0's waits must run into their bound, not spin for ever
            kmap_from_kmers_distributed(kh, cnts, lab, conseqs, K, n_max_iter=6 if rank == 0 else 0, random_seed=SEED, mode=0)
        except RuntimeError as e:
            msg = str(e)
        (Path(out_dir) / f"timeout_rank{rank}.txt").write_text(f"{time.perf_counter() - t0:.2f}|{msg}")
    finally:
        dist.destroy_process_group()


def test_peer_exchange_wait_is_bounded(tmp_path):
    """a rank whose peer never pushes: the apply kernel's wait gives up after its 2-second bound (once: the sticky flag lets the rest
    of the segment through), the host raises -- no kernel spins for ever, the other rank ends normally"""
    import torch.multiprocessing as mp
    mp.spawn(_timeout_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    t0, m0 = (tmp_path / "timeout_rank0.txt").read_text().split("|", 1)
    t1, m1 = (tmp_path / "timeout_rank1.txt").read_text().split("|", 1)
    assert "did not arrive within the wait bound" in m0 and m1 == ""
    assert 1.5 < float(t0) < 30.0


def _verb_timeout_worker(rank, world, port, res_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KMAP_DIST_EXCHANGE="direct", KMAP_PEER_TIMEOUT_MS="1500",
                      WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), KMAP_DIST_BACKEND="gloo", KMAP_DIST_SAME_GPU="1")
    import torch
    import torch.distributed as dist
    import kmap_amd.distributed as D
    import kmap_amd.visualization as V
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if rank == 1:                       # this rank's loop runs no iteration: it never pushes
            real = D.kmap_from_kmers_distributed

            def lazy(*a, **kw):
                kw["n_max_iter"] = 0
                return real(*a, **kw)
            D.kmap_from_kmers_distributed = lazy
        what = "returned"
        try:
            V._visualize_kmers_impl(res_dir, False, None, dist, rank)
        except D.PeerTimeout:
            what = "PeerTimeout"
        (Path(res_dir) / f"verb_rank{rank}.txt").write_text(what)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_peer_timeout_raises_out_of_the_verb_and_writes_no_file(tmp_path):
    """a time-out of the peer-direct exchange inside `visualize_kmers`: the loop driver raises PeerTimeout after the segment (the
    blocks that gave up applied nothing; whatever the others applied is discarded with the exception), and low_dim_data.tsv is never
    written from such coordinates (reference visualization.py:36-87 writes it last)"""
    import pickle
    import torch.multiprocessing as mp
    from kmap_amd._toml import dump_toml
    from kmap_amd.kmer_count import read_default_config_file
    kh, cnts, lab, conseqs = _inputs()
    kh = np.unique(kh).astype(np.uint32)
    lab, cnts = lab[:len(kh)], np.ones(len(kh), np.int64)
    cfg = read_default_config_file()
    cfg["visualization"].update(n_max_iter=8, random_seed=11, gen_fig_flag=False)
    dump_toml(cfg, tmp_path / "config.toml")
    with open(tmp_path / "sample_kmers.pkl", "wb") as fh:
        pickle.dump([kh, cnts, lab, conseqs], fh)
    with open(tmp_path / "sample_kmer_hamdist_mat.pkl", "wb") as fh:
        pickle.dump([K, None, np.repeat(lab, cnts)], fh)
    mp.spawn(_verb_timeout_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "verb_rank0.txt").read_text() == "PeerTimeout"
    assert (tmp_path / "verb_rank1.txt").read_text() == "returned"
    assert not (tmp_path / "low_dim_data.tsv").exists()


def _bad_handle_worker(rank, world, port, out_dir, exchange):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import warnings
    import torch
    import torch.distributed as dist
    import kmap_amd.distributed as D
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        real = D.PeerExchange

        class Broken(real):                 # rank 0 receives a handle of rank 1 that maps nothing
            def __init__(self, session, n, dist_, group=None):
                super().__init__(session, n, dist_, group, _corrupt_handle_of=1 if dist_.get_rank() == 0 else None)
        D.PeerExchange = Broken
        kh, cnts, lab, conseqs = _inputs()
        tr, what = {}, ""
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            try:
                D.kmap_from_kmers_distributed(kh, cnts, lab, conseqs, K, n_max_iter=ITERS, random_seed=SEED, mode=1, trace=tr, exchange=exchange)
                what = "ran:" + tr["exchange"] + ":" + ("warned" if any("falling back" in str(w.message) for w in caught) else "silent")
            except D.PeerExchangeError as e:
                what = "PeerExchangeError:" + str(e)
        np.savez(Path(out_dir) / f"bad_{exchange}_rank{rank}.npz", what=what, losses=tr.get("losses", np.zeros(0)))
    finally:
        dist.destroy_process_group()


def test_peer_exchange_setup_failure_is_seen_by_every_rank_and_auto_falls_back(tmp_path):
    """fault injection: one rank's copy of a peer's IPC handle is corrupt.  KMAP_DIST_EXCHANGE=direct: BOTH ranks raise
    PeerExchangeError before any iteration (the rank that could map everything learns of the failure through the all-reduced success
    flag instead of waiting for a peer that has left); =auto: both fall back to one all-reduce per iteration, warn once, and
    produce the all-reduce run's losses."""
    import torch.multiprocessing as mp
    for exchange in ("direct", "auto"):
        mp.spawn(_bad_handle_worker, args=(2, _free_port(), str(tmp_path), exchange), nprocs=2, join=True)
    d0, d1 = (str(np.load(tmp_path / f"bad_direct_rank{r}.npz")["what"]) for r in range(2))
    assert d0.startswith("PeerExchangeError:") and "this rank" in d0 and "cannot be mapped" in d0
    assert d1.startswith("PeerExchangeError:") and "another rank" in d1
    a = [np.load(tmp_path / f"bad_auto_rank{r}.npz") for r in range(2)]
    assert all(str(x["what"]) == "ran:all_reduce:warned" for x in a)
    np.testing.assert_array_equal(a[0]["losses"], a[1]["losses"])
    assert len(a[0]["losses"]) == ITERS


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("exchange", ["rccl", "direct"])
@pytest.mark.parametrize("mode", [1, 0])   # SEQ, FAST
def test_two_ranks_one_gpu_equals_single(tmp_path, mode, exchange):
    """exchange = direct (KMAP_DIST_EXCHANGE): the iteration message travels by peer-to-peer stores into IPC-mapped receive areas +
    flags instead of an all-reduce (PeerExchange; two processes on one GPU open each other's handles) -- the same bits."""
    import torch.multiprocessing as mp
    import kmap_amd.visualization as V
    mp.spawn(_worker, args=(2, _free_port(), mode, str(tmp_path), exchange), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / f"m{mode}_rank0.npz"), np.load(tmp_path / f"m{mode}_rank1.npz")
    np.testing.assert_array_equal(r0["last"], r1["last"])
    np.testing.assert_array_equal(r0["losses"], r1["losses"])
    assert int(r0["d_rows"]) == 751 and int(r1["d_rows"]) == 750      # each rank computed only its rows of D (N^2 / G bytes)
    kh, cnts, lab, conseqs = _inputs()
    tr = {}
    best, _ = V.kmap_from_kmers(kh, cnts, lab, conseqs, K, n_max_iter=ITERS, random_seed=SEED, mode=mode, trace=tr)
    np.testing.assert_allclose(r0["losses"], tr["losses"], rtol=1e-6)
    # per-row sums do not depend on the sharding (a row is always summed by one lane / one wave)
    np.testing.assert_array_equal(r0["last"], tr["last_coords"])
    np.testing.assert_array_equal(r0["best"], best)


def test_two_ranks_with_repeated_kmers_equal_single(tmp_path, monkeypatch):
    """a sample with counts of 1 .. 6: every rank stores the runs of equal sums rows of its block once and reads them through the
    row map (dedupe_sums_rows in kmap_from_kmers_distributed) -- the same coordinates as the single-GPU run, which does the same on
    all rows, and the same as the single-GPU run on the expanded matrix."""
    import torch.multiprocessing as mp
    import kmap_amd.visualization as V
    monkeypatch.setenv("KMAP_TEST_REPEATS", "1")
    mp.spawn(_worker, args=(2, _free_port(), 1, str(tmp_path), "rccl"), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "m1_rank0.npz"), np.load(tmp_path / "m1_rank1.npz")
    np.testing.assert_array_equal(r0["last"], r1["last"])
    kh, cnts, lab, conseqs = _inputs()
    assert int(cnts.sum()) > 2 * N
    tr = {}
    best, _ = V.kmap_from_kmers(kh, cnts, lab, conseqs, K, n_max_iter=ITERS, random_seed=SEED, mode=1, trace=tr)
    np.testing.assert_array_equal(r0["last"], tr["last_coords"])
    np.testing.assert_array_equal(r0["best"], best)
    real = V.dedupe_sums_rows
    monkeypatch.setattr(V, "dedupe_sums_rows", lambda sums_d, nrows, lds, **kw: (sums_d, None, nrows))     # the expanded matrix
    tr2 = {}
    best2, _ = V.kmap_from_kmers(kh, cnts, lab, conseqs, K, n_max_iter=ITERS, random_seed=SEED, mode=1, trace=tr2)
    monkeypatch.setattr(V, "dedupe_sums_rows", real)
    np.testing.assert_array_equal(tr2["last_coords"], tr["last_coords"])
    np.testing.assert_array_equal(best2, best)


def _count_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import pickle
    import torch
    import torch.distributed as dist
    from kmap_amd import synth
    from kmap_amd.distributed import make_dist_device_seq
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash
    from kmap_amd.motif_discovery import find_motif
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq, borders = synth.synth_reads(30011, 75, 4)          # odd read count: uneven shards
        ds = make_dist_device_seq(seq, borders, dist)
        dc = DeviceCounts()
        out = {"shard": (ds.first_read, ds.n_local_reads)}
        for k in (6, 9, 11):
            for dedupe in (True, False):
                ds.count(dc, k, dedupe=dedupe, merge_revcom=True)
                out[(k, dedupe)] = dc.fetch()
        out["scan"] = ds.scan(8, kmer2hash("ATCGATAG"), 2, True)
        mdef = type("M", (), dict(max_ham_dist=2, p_uniform=0.004241943, ratio_mu=1.0, ratio_std=0.04864974, ratio_cutoff=1.3094))()
        ds.reset()
        r = find_motif(None, 8, mdef.max_ham_dist, mdef.p_uniform, mdef.ratio_mu, mdef.ratio_std, mdef.ratio_cutoff,
                       save_kmer_cnt_flag=False, dev_seq=ds)
        out["motifs"] = {int(h): v for h, v in r.items()}
        with open(Path(out_dir) / f"cnt_rank{rank}.pkl", "wb") as fh:
            pickle.dump(out, fh)
        dc.close()
        ds.close()
    finally:
        dist.destroy_process_group()


def _range_inputs(poly=True):
    from kmap_amd import synth
    seq, borders = synth.synth_reads(20011, 75, 9)
    for r in range(0, len(borders), 41):                     # palindromic repeats (rc(ACGTACGTACGT) = itself), poly-A / poly-T pairs
        st, en = borders[r]
        if r % 3 == 0 or not poly:
            seq[st:en] = np.resize(np.array([0, 1, 2, 3], np.uint8), en - st)
        else:
            seq[st:en] = 0 if r % 3 == 1 else 3
    return seq, borders


def _range_count_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import pickle
    import torch
    import torch.distributed as dist
    from kmap_amd.distributed import make_dist_device_seq
    from kmap_amd.kmer_count import DeviceCounts
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq, borders = _range_inputs()
        ds = make_dist_device_seq(seq, borders, dist, shard_counts=True)
        dc = DeviceCounts()
        out = {}
        for k in (11, 12, 15):
            for dedupe in (True, False):
                for merge in (True, False):
                    ds.count(dc, k, dedupe=dedupe, merge_revcom=merge)
                    out[(k, dedupe, merge)] = dc.fetch()
        out["top"] = dc.topk(5)                               # the adopted table serves the handle's other queries
        with open(Path(out_dir) / f"range_rank{rank}.pkl", "wb") as fh:
            pickle.dump(out, fh)
        dc.close()
        ds.close()
    finally:
        dist.destroy_process_group()


def test_key_range_sharded_counting_three_ranks(tmp_path):
    """Bins owned by key range (three ranks: unequal slices): local table -> all-reduced presence nibbles -> local revcom merge ->
    one SUM-reduce per slice -> compaction of the own slice -> all-gather of the shards == the single-process count (oracle),
    arrays and dtypes, with and without per-read dedupe and revcom merge, k = 11 / 12 (palindromes) / 15 (4-GiB table)."""
    import pickle
    import torch.multiprocessing as mp
    from oracle import oracle as O
    mp.spawn(_range_count_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    res = [pickle.load(open(tmp_path / f"range_rank{r}.pkl", "rb")) for r in range(3)]
    seq, borders = _range_inputs()
    for k in (11, 12, 15):
        for dedupe in (True, False):
            for merge in (True, False):
                ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=merge)
                for r in res:
                    u, c = r[(k, dedupe, merge)]
                    np.testing.assert_array_equal(u, ou)
                    np.testing.assert_array_equal(c, oc)
                    assert u.dtype == ou.dtype and c.dtype == oc.dtype
    assert res[0]["top"][0].tolist() == res[1]["top"][0].tolist() == res[2]["top"][0].tolist()


def _kept_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import pickle
    import torch
    import torch.distributed as dist
    import kmap_amd.motif_discovery as md
    from kmap_amd.distributed import make_dist_device_seq
    from kmap_amd.kmer_count import DeviceCounts
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        md.TOPK_DEVICE_MIN = 1000                            # find_motif takes its candidates from the device top-k above this size
        seq, borders = _range_inputs()
        ds = make_dist_device_seq(seq, borders, dist, shard_counts=True)
        ds.keep_sharded, ds.full_table_rank = True, 0
        dc = DeviceCounts()
        out = {}
        cands = {13: [0, 5, 0x1B1B1B1, 4 ** 13 - 1], 16: [0, 0x1B1B1B1B, 4 ** 16 - 1]}
        for k in (13, 16):                                   # 16: int64 keys, 16-GiB tables
            ds.count(dc, k, dedupe=True, merge_revcom=True)          # a masked re-count of find_motif: nobody gathers the table
            assert dc._shard is not None and dc._full is None and dc.fetch_full() is None
            ds.count(dc, k, dedupe=True, merge_revcom=True, gather_full=True)   # the table k{k}.pkl is written from
            assert dc._shard is not None and (dc._full is not None) == (rank == 0)
            out[k] = {"n_uniq": dc.n_uniq, "n_local": dc._shard.n_local, "total": dc.total(), "top": dc.topk(7),
                      "mass": dc.hamball_mass(np.array(cands[k], np.uint64), 3, True)}
            out[k]["all"] = dc.fetch()                               # collective: EVERY rank calls it, with full_table_rank set
            out[k]["total_after"] = dc.total()                       # ... and the next collective still pairs up
            if rank == 0:
                out[k]["full"] = dc.fetch_full()
        ds.reset()
        # the reads hold poly-A / poly-T stretches: a consensus within the radius of the all-T k-mer also matches the windows that touch a
        # separator (the reference's all-ones invalid hash, kmer_count.py:580-610) and masks k - 1 positions into the NEXT read --
        # across a shard boundary too (DistDeviceSeq.mask)
        r = md.find_motif(None, 13, 3, 3.0e-6, 1.0, 0.05, 1.3, top_k=5, n_trial=3, save_kmer_cnt_flag=False, dev_seq=ds)
        out["motifs"] = {int(h): v for h, v in r.items()}
        with open(Path(out_dir) / f"kept_rank{rank}.pkl", "wb") as fh:
            pickle.dump(out, fh)
        dc.close()
        ds.close()
    finally:
        dist.destroy_process_group()


def test_find_motif_on_counts_that_stay_sharded(tmp_path):
    """Key-range counting with the table LEFT sharded (three ranks; distributed.CountShard): no rank but the k{k}.pkl writer receives
    the (k-mer, count) list; n_uniq / total / top-k (count descending, index in the whole table ascending) / Hamming-ball masses are
    local partials + one tiny collective each and equal the oracle's on the whole table; find_motif on such a table == find_motif of
    one GPU on all reads (same device top-k rule) on every rank.  k = 13 and k = 16 (int64 keys; reference motif_discovery.py:655-701)."""
    import pickle
    import torch.multiprocessing as mp
    import kmap_amd.motif_discovery as md
    from kmap_amd.motif_discovery import DeviceSeq
    from oracle import oracle as O
    mp.spawn(_kept_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    res = [pickle.load(open(tmp_path / f"kept_rank{r}.pkl", "rb")) for r in range(3)]
    seq, borders = _range_inputs()
    cands = {13: [0, 5, 0x1B1B1B1, 4 ** 13 - 1], 16: [0, 0x1B1B1B1B, 4 ** 16 - 1]}
    for k in (13, 16):
        ou, oc = O.count_kmers(seq, borders, k, rep_mode=False, revcom_mode=True)
        order = np.lexsort((np.arange(len(oc)), -oc.astype(np.int64)))[:7]
        mass = O.hamball_mass(ou, oc, k, np.array(cands[k], np.uint64), 3, True)
        assert sum(r[k]["n_local"] for r in res) == len(ou) and min(r[k]["n_local"] for r in res) > 0
        for r in res:
            assert r[k]["n_uniq"] == len(ou) and r[k]["total"] == int(oc.sum())
            idx, kh, cnt = r[k]["top"]
            assert idx.tolist() == order.tolist() and kh.tolist() == ou[order].tolist() and cnt.tolist() == oc[order].tolist()
            assert kh.dtype == ou.dtype
            np.testing.assert_array_equal(r[k]["mass"], mass)
        fu, fc = res[0][k]["full"]                           # the writer rank holds the gathered table as well
        np.testing.assert_array_equal(fu, ou)
        np.testing.assert_array_equal(fc, oc)
        assert fu.dtype == ou.dtype and fc.dtype == oc.dtype
        for r in res:                                        # fetch() on a sharded table: collective, same answer on every rank
            np.testing.assert_array_equal(r[k]["all"][0], ou)
            np.testing.assert_array_equal(r[k]["all"][1], oc)
            assert r[k]["total_after"] == int(oc.sum())
    old = md.TOPK_DEVICE_MIN
    md.TOPK_DEVICE_MIN = 1000
    try:
        ds = DeviceSeq(seq, borders)
        single = md.find_motif(None, 13, 3, 3.0e-6, 1.0, 0.05, 1.3, top_k=5, n_trial=3, save_kmer_cnt_flag=False, dev_seq=ds)
        ds.close()
    finally:
        md.TOPK_DEVICE_MIN = old
    single = {int(h): v for h, v in single.items()}
    assert len(single) >= 1
    for r in res:
        assert list(r["motifs"]) == list(single)
        for h in single:
            assert r["motifs"][h] == single[h]


def _keyspace_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import pickle
    import torch
    import torch.distributed as dist
    import kmap_amd.motif_discovery as md
    from kmap_amd import distributed as D
    from kmap_amd.kmer_count import DeviceCounts
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq, borders = _range_inputs()
        out = {}
        # a spy on the collectives: key-space counting must not move a single table byte (sizes and, for small tables, the
        # (k-mer, count) shards are all that is exchanged)
        moved = {"all_reduce": 0, "reduce": 0}
        real_ar, real_red = dist.all_reduce, dist.reduce

        def spy_ar(t, *a, **kw):
            moved["all_reduce"] += t.numel() * t.element_size()
            return real_ar(t, *a, **kw)

        def spy_red(t, *a, **kw):
            moved["reduce"] += t.numel() * t.element_size()
            return real_red(t, *a, **kw)
        dist.all_reduce, dist.reduce = spy_ar, spy_red
        ds = D.make_dist_device_seq(seq, borders, dist, key_space=True)       # forced: k = 11 lies below KEY_SPACE_MIN_K
        dc = DeviceCounts()
        for k in (11, 14, 16):
            for dedupe in (True, False):
                for merge in (True, False):
                    ds.count(dc, k, dedupe=dedupe, merge_revcom=merge)
                    out[(k, dedupe, merge)] = dc.fetch()
        out["moved"] = dict(moved)
        dist.all_reduce, dist.reduce = real_ar, real_red
        # the default rule: k >= KEY_SPACE_MIN_K counts by key space, smaller k by read shards + all-reduce; masks reach both copies
        ds.close()
        md.TOPK_DEVICE_MIN = 1000
        ds = D.make_dist_device_seq(seq, borders, dist)
        ds.keep_sharded = True
        ds.reset()
        r13 = md.find_motif(None, 13, 3, 3.0e-6, 1.0, 0.05, 1.3, top_k=5, n_trial=3, save_kmer_cnt_flag=False, dev_seq=ds)
        assert ds._full is not None and ds._done_full > 0 and ds._done_shard == 0        # k = 13 never masked the shard
        out["motifs13"] = {int(h): v for h, v in r13.items()}
        ds.reset()
        r8 = md.find_motif(None, 8, 2, 0.004241943, 1.0, 0.04864974, 1.3094, save_kmer_cnt_flag=False, dev_seq=ds)
        assert ds._done_full == 0                                                          # k = 8 never masked the full copy
        out["motifs8"] = {int(h): v for h, v in r8.items()}
        ds.reset()
        # a mask applied before the full copy exists is replayed onto it; the working reads agree
        ds2 = D.make_dist_device_seq(seq, borders, dist)
        cons = np.array([0x1B1B1B1, 4 ** 13 - 1], np.uint64)
        ds2.mask(13, cons, np.array([2, 1], np.int32))
        ds2.count(dc, 13, dedupe=False, merge_revcom=True)
        out["masked13"] = dc.fetch()
        # KMAP_DIST_KEYSPACE overrides the rule: 0 = read shards + table all-reduce at every k, 1 = key space from k = 11 on
        ds2.reset()
        os.environ["KMAP_DIST_KEYSPACE"] = "0"
        ds2.count(dc, 14, dedupe=False, merge_revcom=True)
        out["k14_allreduce_form"] = dc.fetch()
        os.environ["KMAP_DIST_KEYSPACE"] = "1"
        ds2.count(dc, 11, dedupe=True, merge_revcom=True)
        out["k11_keyspace_by_switch"] = dc.fetch()
        os.environ.pop("KMAP_DIST_KEYSPACE")
        with open(Path(out_dir) / f"ks_rank{rank}.pkl", "wb") as fh:
            pickle.dump(out, fh)
        dc.close()
        ds.close()
        ds2.close()
    finally:
        dist.destroy_process_group()


def test_key_space_counting_three_ranks(tmp_path):
    """counting by KEY SPACE under a process group (three ranks sharing the test box's GPU): every rank holds all reads and
    computes only its key range (kmap_counts_run_packed_range_dev) -- the gathered tables equal the oracle's single-process count bit for
    bit at k = 11, 14, 16 with and without per-read dedupe and revcom merge, with ZERO table bytes through all_reduce / reduce; the
    default rule takes key space from k = 13 on, find_motif on a table that stays sharded == one GPU on all reads, masks applied before
    and after the full copy exists reach it (reference kmer_count.py:476-491,580-610,643-685)."""
    import pickle
    import torch.multiprocessing as mp
    import kmap_amd.motif_discovery as md
    from kmap_amd.kmer_count import DeviceCounts
    from kmap_amd.motif_discovery import DeviceSeq
    from oracle import oracle as O
    mp.spawn(_keyspace_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    res = [pickle.load(open(tmp_path / f"ks_rank{r}.pkl", "rb")) for r in range(3)]
    seq, borders = _range_inputs()
    for k in (11, 14, 16):
        for dedupe in (True, False):
            for merge in (True, False):
                ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=merge)
                for r in res:
                    u, c = r[(k, dedupe, merge)]
                    np.testing.assert_array_equal(u, ou)
                    np.testing.assert_array_equal(c, oc)
                    assert u.dtype == ou.dtype and c.dtype == oc.dtype
    for r in res:                # 12 count passes: no all-reduce / reduce of anything table-sized (the 16-GiB table of k = 16 included)
        assert r["moved"]["reduce"] == 0 and r["moved"]["all_reduce"] < 4096, r["moved"]
    old = md.TOPK_DEVICE_MIN
    md.TOPK_DEVICE_MIN = 1000
    try:
        ds, dc = DeviceSeq(seq, borders), DeviceCounts()
        s13 = md.find_motif(None, 13, 3, 3.0e-6, 1.0, 0.05, 1.3, top_k=5, n_trial=3, save_kmer_cnt_flag=False, dev_seq=ds)
        ds.reset()
        s8 = md.find_motif(None, 8, 2, 0.004241943, 1.0, 0.04864974, 1.3094, save_kmer_cnt_flag=False, dev_seq=ds)
        ds.reset()
        ds.mask(13, np.array([0x1B1B1B1, 4 ** 13 - 1], np.uint64), np.array([2, 1], np.int32))
        ds.count(dc, 13, dedupe=False, merge_revcom=True)
        mu, mc = dc.fetch()
        dc.close()
        ds.close()
    finally:
        md.TOPK_DEVICE_MIN = old
    assert len(s13) >= 1 and len(s8) >= 1
    for r in res:
        assert r["motifs13"] == {int(h): v for h, v in s13.items()}
        assert r["motifs8"] == {int(h): v for h, v in s8.items()}
        np.testing.assert_array_equal(r["masked13"][0], mu)
        np.testing.assert_array_equal(r["masked13"][1], mc)
        for key, ref in (("k14_allreduce_form", (14, False, True)), ("k11_keyspace_by_switch", (11, True, True))):
            np.testing.assert_array_equal(r[key][0], r[ref][0])
            np.testing.assert_array_equal(r[key][1], r[ref][1])


def test_read_sharded_counting_scan_and_find_motif(tmp_path):
    """Reads sharded over two ranks (histogram all-reduce): counts, scan hits and find_motif decisions equal the
    single-GPU run and the CPU oracle."""
    import pickle
    import torch.multiprocessing as mp
    from kmap_amd import synth
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash
    from kmap_amd.motif_discovery import DeviceSeq, find_motif
    from oracle import oracle as O
    mp.spawn(_count_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = [pickle.load(open(tmp_path / f"cnt_rank{r}.pkl", "rb")) for r in range(2)]
    assert res[0]["shard"] == (0, 15006) and res[1]["shard"] == (15006, 15005)
    seq, borders = synth.synth_reads(30011, 75, 4)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    for k in (6, 9, 11):
        for dedupe in (True, False):
            ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=True)
            for r in res:
                np.testing.assert_array_equal(r[(k, dedupe)][0], ou)
                np.testing.assert_array_equal(r[(k, dedupe)][1], oc)
    hits, pos = ds.scan(8, kmer2hash("ATCGATAG"), 2, True)
    for r in res:
        np.testing.assert_array_equal(r["scan"][0], hits)
        np.testing.assert_array_equal(r["scan"][1], pos)
    mdef = dict(max_ham_dist=2, p_uniform=0.004241943, ratio_mu=1.0, ratio_std=0.04864974, ratio_cutoff=1.3094)
    single = find_motif(None, 8, mdef["max_ham_dist"], mdef["p_uniform"], mdef["ratio_mu"], mdef["ratio_std"], mdef["ratio_cutoff"],
                        save_kmer_cnt_flag=False, dev_seq=ds)
    assert len(single) >= 2
    for r in res:
        assert r["motifs"] == {int(h): v for h, v in single.items()}
    dc.close()
    ds.close()


def test_visualize_kmers_cli_under_torchrun(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m kmap_amd visualize_kmers` (compact hand-off, rows sharded over
    two ranks that share the test box's GPU, gloo) writes the same low_dim_data.tsv as the single-process verb."""
    import pickle
    import shutil
    import subprocess
    from kmap_amd._toml import dump_toml
    from kmap_amd.kmer_count import read_default_config_file
    kh, cnts, lab, conseqs = _inputs()
    kh = np.unique(kh).astype(np.uint32)
    lab, cnts = lab[:len(kh)], np.ones(len(kh), np.int64)
    cnts[::7] = 3                                               # expanded N > number of unique k-mers
    outs = []
    for tag in ("single", "dist", "dist_auto"):       # dist_auto: KMAP_DIST_EXCHANGE=auto -- the validated peer-direct exchange inside the verb
        res = tmp_path / tag
        res.mkdir()
        cfg = read_default_config_file()
        cfg["visualization"].update(n_max_iter=15, random_seed=11, gen_fig_flag=False)
        dump_toml(cfg, res / "config.toml")
        with open(res / "sample_kmers.pkl", "wb") as fh:
            pickle.dump([kh, cnts, lab, conseqs], fh)
        with open(res / "sample_kmer_hamdist_mat.pkl", "wb") as fh:
            pickle.dump([K, None, np.repeat(lab, cnts)], fh)
        env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if tag == "single":
            cmd = [sys.executable, "-m", "kmap_amd", "visualize_kmers", "--res_dir", str(res)]
        else:
            env.update(KMAP_DIST_BACKEND="gloo", KMAP_DIST_SAME_GPU="1")
            if tag == "dist_auto":
                env.update(KMAP_DIST_EXCHANGE="auto")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), "-m", "kmap_amd", "visualize_kmers", "--res_dir", str(res)]
        r = subprocess.run(cmd, env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((res / "low_dim_data.tsv").read_text())
    assert outs[0] == outs[1] == outs[2] and outs[0].count("\n") == int(cnts.sum()) + 1


def _cyclic_worker(rank, world, port, out_dir, n, iters, exchange="rccl"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KMAP_DIST_EXCHANGE=exchange)
    import torch
    import torch.distributed as dist
    import kmap_amd.visualization as V
    from kmap_amd.distributed import kmap_from_kmers_distributed
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(3)
        kh = rng.integers(0, 4 ** K, size=n, dtype=np.uint64)
        lab = np.sort(rng.integers(0, 3, size=n)).astype(np.int64)
        tr = {}
        best, _ = kmap_from_kmers_distributed(kh, np.ones(n, np.int64), lab, ["ACGTACGT", "ACGTAC"], K, n_max_iter=iters,
                                              random_seed=SEED, mode=V.EMBED_FAST, trace=tr)
        np.savez(Path(out_dir) / f"cyc_rank{rank}.npz", best=best, last=tr["last_coords"], losses=tr["losses"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cyclic,exchange", [("1", "rccl"), ("0", "rccl"), ("1", "direct")])
def test_cyclic_symmetric_shards_match_single_gpu(tmp_path, monkeypatch, cyclic, exchange):
    """FAST at N >= 16384 under torch.distributed: the ranks own cyclic 256-row blocks and evaluate each unordered pair once;
    the all-reduced gradient equals the single-GPU symmetric kernel's up to the order of the partial sums.  KMAP_DIST_CYCLIC=0:
    the same run on contiguous row blocks (the row-wise FAST kernel, every ordered pair) -- the layout below N = 16 384."""
    import torch.multiprocessing as mp
    import kmap_amd.visualization as V
    monkeypatch.setenv("KMAP_DIST_CYCLIC", cyclic)  # inherited by the spawned ranks
    n, iters = 16384 + 3 * 256 + 77, 6            # 68 row blocks, the last one ragged; odd split over 3 ranks
    mp.spawn(_cyclic_worker, args=(3, _free_port(), str(tmp_path), n, iters, exchange), nprocs=3, join=True)
    r = [np.load(tmp_path / f"cyc_rank{i}.npz") for i in range(3)]
    for i in (1, 2):                              # identical state machines on every rank
        np.testing.assert_array_equal(r[0]["last"], r[i]["last"])
        np.testing.assert_array_equal(r[0]["losses"], r[i]["losses"])
    rng = np.random.default_rng(3)
    kh = rng.integers(0, 4 ** K, size=n, dtype=np.uint64)
    lab = np.sort(rng.integers(0, 3, size=n)).astype(np.int64)
    tr = {}
    V.kmap_from_kmers(kh, np.ones(n, np.int64), lab, ["ACGTACGT", "ACGTAC"], K, n_max_iter=iters, random_seed=SEED,
                      mode=V.EMBED_FAST, trace=tr)
    # (the row-wise kernel takes one log per 8 (1 - q) factors in another grouping than the tile kernel: a systematic ~3e-6)
    np.testing.assert_allclose(r[0]["losses"], tr["losses"], rtol=2e-6 if cyclic == "1" else 6e-6)
    # FAST sums are order-dependent in the last bits and the first steps from a random start are violent (the loss falls
    # 100x in one step), so a handful of coordinates drift to ~1e-4 of the embedding's extent within 6 iterations
    scale = np.abs(tr["last_coords"]).max()
    diff = np.abs(r[0]["last"] - tr["last_coords"])
    assert diff.max() <= 1e-3 * scale and np.quantile(diff, 0.99) <= 2e-5 * scale      # (5.2e-4 with the numpy neighbour choice this N now takes)


def _seq_shard_worker(rank, world, port, out_dir, n, iters):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from kmap_amd.distributed import kmap_from_kmers_distributed
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        kh = rng.integers(0, 4 ** K, size=n, dtype=np.uint64)
        lab = np.sort(rng.integers(0, 3, size=n)).astype(np.int64)
        tr = {}
        best, _ = kmap_from_kmers_distributed(kh, np.ones(n, np.int64), lab, ["ACGTACGT", "ACGTAC"], K, n_max_iter=iters, random_seed=SEED, mode=1, trace=tr)
        np.savez(Path(out_dir) / f"seqsh_rank{rank}.npz", best=best, last=tr["last_coords"], losses=tr["losses"])
    finally:
        dist.destroy_process_group()


def test_seq_row_shards_in_the_producer_adder_form_equal_single_gpu(tmp_path):
    """SEQ (the package default) at an N where a rank's share runs in the producer / adder form of the force kernel (N >= 3072,
    fewer rows than 2.5 rounds of quad waves; embed_seq.hip): three ranks with contiguous row blocks -- a row is summed by one adder
    lane in column order whatever the sharding, so coordinates and losses equal the single-GPU run's, bit for bit, on every rank."""
    import torch.multiprocessing as mp
    import kmap_amd.visualization as V
    n, iters = 4099, 9                            # odd N (last chunk ragged), rows 1367 / 1366 / 1366
    mp.spawn(_seq_shard_worker, args=(3, _free_port(), str(tmp_path), n, iters), nprocs=3, join=True)
    r = [np.load(tmp_path / f"seqsh_rank{i}.npz") for i in range(3)]
    rng = np.random.default_rng(5)
    kh = rng.integers(0, 4 ** K, size=n, dtype=np.uint64)
    lab = np.sort(rng.integers(0, 3, size=n)).astype(np.int64)
    tr = {}
    best, _ = V.kmap_from_kmers(kh, np.ones(n, np.int64), lab, ["ACGTACGT", "ACGTAC"], K, n_max_iter=iters, random_seed=SEED, mode=V.EMBED_SEQ, trace=tr)
    for x in r:
        np.testing.assert_array_equal(x["last"], tr["last_coords"])
        np.testing.assert_array_equal(x["best"], best)
        np.testing.assert_allclose(x["losses"], tr["losses"], rtol=1e-6)


def _seed_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from kmap_amd.distributed import kmap_from_kmers_distributed
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kh, cnts, lab, conseqs = _inputs()
        tr = {}
        best, _ = kmap_from_kmers_distributed(kh, cnts, lab, conseqs, K, n_max_iter=8, random_seed=None, mode=1, trace=tr)
        np.savez(Path(out_dir) / f"seed_rank{rank}.npz", best=best, last=tr["last_coords"], losses=tr["losses"], seed=tr["seed"])
    finally:
        dist.destroy_process_group()


def test_default_seed_is_shared_by_all_ranks(tmp_path):
    """random_seed = "default" (None: OS entropy in the reference): rank 0 draws the seed and broadcasts it, so every rank
    starts from the same coordinates / placeholders / jitter stream and the state machines stay identical."""
    import torch.multiprocessing as mp
    mp.spawn(_seed_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "seed_rank0.npz"), np.load(tmp_path / "seed_rank1.npz")
    assert int(r0["seed"]) == int(r1["seed"])
    np.testing.assert_array_equal(r0["last"], r1["last"])
    np.testing.assert_array_equal(r0["best"], r1["best"])
    np.testing.assert_array_equal(r0["losses"], r1["losses"])
    assert np.isfinite(r0["losses"]).all() and len(r0["losses"]) == 8


def _occ_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from kmap_amd import synth
    from kmap_amd.distributed import make_dist_device_seq
    from kmap_amd.kmer_count import _pkg_file, init_motif_def_dict
    from kmap_amd.motif_discovery import gen_motif_occurence_file
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq, borders = synth.synth_reads(20011, 90, 6)
        ds = make_dist_device_seq(seq, borders, dist)
        mdd = init_motif_def_dict(_pkg_file("default_motif_def_table.csv"))
        np.random.seed(3)
        gen_motif_occurence_file(["AATCGATAGC", "CCTACGTA", "AAAAAA"], mdd, None, Path(out_dir) / f"occ_rank{rank}.csv", True,
                                 dev_seq=ds, write=(rank == 0))
        ds.close()
    finally:
        dist.destroy_process_group()


def test_occurrence_csv_from_read_sharded_seq(tmp_path):
    """gen_motif_occurence_file on a read-sharded DistDeviceSeq (3 ranks, uneven shards) writes the single-GPU file: the
    gathered hits are paired with the GLOBAL read count / read lengths."""
    import torch.multiprocessing as mp
    from kmap_amd import synth
    from kmap_amd.kmer_count import _pkg_file, init_motif_def_dict
    from kmap_amd.motif_discovery import DeviceSeq, gen_motif_occurence_file
    mp.spawn(_occ_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    assert not (tmp_path / "occ_rank1.csv").exists() and not (tmp_path / "occ_rank2.csv").exists()
    seq, borders = synth.synth_reads(20011, 90, 6)
    ds = DeviceSeq(seq, borders)
    np.random.seed(3)
    gen_motif_occurence_file(["AATCGATAGC", "CCTACGTA", "AAAAAA"], init_motif_def_dict(_pkg_file("default_motif_def_table.csv")),
                             None, tmp_path / "single.csv", True, dev_seq=ds)
    ds.close()
    got, want = (tmp_path / "occ_rank0.csv").read_text(), (tmp_path / "single.csv").read_text()
    assert got == want and got.count("\n") > 5000


def test_scan_motif_cli_under_torchrun(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m kmap_amd scan_motif` (reads sharded over two ranks that share
    the test box's GPU, gloo): every output file equals the single-process verb's, a second (cached) run re-uses them."""
    import pickle
    import subprocess
    from kmap_amd import synth
    seq, borders = synth.synth_reads(40_001, 60, 12)
    outs = {}
    for tag in ("single", "dist", "dist_auto"):       # dist_auto: KMAP_DIST_EXCHANGE=auto -- the validated peer-direct exchange inside the verb
        res = tmp_path / tag
        res.mkdir()
        over = {"kmer_count": {"min_k": 6, "max_k": 9},
                "motif_discovery": {"motif_pos_density_flag": False, "motif_co_occurence_flag": False, "gen_hamball_flag": False,
                                    "n_total_sample": 400, "n_motif_sample": 200},
                "visualization": {"gen_fig_flag": False, "random_seed": 7, "n_max_iter": 10}}
        synth.write_res_dir(res, seq, borders, over)
        env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
        seeded = "import sys, numpy as np; np.random.seed(123); from kmap_amd.motif_discovery import _scan_motif; _scan_motif(sys.argv[1])"
        if tag == "single":
            cmd = [sys.executable, "-c", seeded, str(res)]
        else:
            env.update(KMAP_DIST_BACKEND="gloo", KMAP_DIST_SAME_GPU="1")
            (tmp_path / "seeded_scan.py").write_text(seeded + "\n")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), str(tmp_path / "seeded_scan.py"), str(res)]
        for attempt in range(2 if tag == "dist" else 1):       # second pass: every branch takes its "already exist" path
            r = subprocess.run(cmd, env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-3000:]
        files = {}
        for f in sorted(res.rglob("*")):
            if f.is_file() and f.name not in ("input.bin.pkl", "input.seqboarder.bin.pkl", "config.toml"):
                files[str(f.relative_to(res))] = f.read_bytes()
        outs[tag] = files
    assert sorted(outs["single"]) == sorted(outs["dist"]) and len(outs["single"]) >= 12
    for name, blob in outs["single"].items():
        if name.endswith(".pkl"):
            a, b = pickle.loads(blob), pickle.loads(outs["dist"][name])
            for x, y in zip(a, b):
                np.testing.assert_array_equal(np.asarray(x, dtype=object) if isinstance(x, list) else x,
                                              np.asarray(y, dtype=object) if isinstance(y, list) else y, err_msg=name)
        else:
            assert blob == outs["dist"][name], name
    assert len((tmp_path / "dist" / "final_conseq.txt").read_text().split()) >= 1


def test_scan_motif_cli_key_space_under_torchrun(tmp_path):
    """`scan_motif` for k = 12..14 under two ranks: k = 12 counts read shards + a table all-reduce, k = 13 / 14 count by KEY SPACE (every
    rank holds all reads and computes its half of the table; the masked re-counts of find_motif's later rounds replay the masks on the
    full copy) -- every output file, k{k}.pkl included, equals the single-process verb's."""
    import pickle
    import subprocess
    from kmap_amd import synth
    seq, borders = synth.synth_reads(40_001, 60, 12)
    outs = {}
    for tag in ("single", "dist"):
        res = tmp_path / tag
        res.mkdir()
        over = {"kmer_count": {"min_k": 12, "max_k": 14},
                "motif_discovery": {"motif_pos_density_flag": False, "motif_co_occurence_flag": False, "gen_hamball_flag": False,
                                    "n_total_sample": 400, "n_motif_sample": 200},
                "visualization": {"gen_fig_flag": False, "random_seed": 7, "n_max_iter": 10}}
        synth.write_res_dir(res, seq, borders, over)
        env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
        seeded = "import sys, numpy as np; np.random.seed(123); from kmap_amd.motif_discovery import _scan_motif; _scan_motif(sys.argv[1])"
        if tag == "single":
            cmd = [sys.executable, "-c", seeded, str(res)]
        else:
            env.update(KMAP_DIST_BACKEND="gloo", KMAP_DIST_SAME_GPU="1")
            (tmp_path / "seeded_scan.py").write_text(seeded + "\n")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), str(tmp_path / "seeded_scan.py"), str(res)]
        r = subprocess.run(cmd, env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[tag] = {str(f.relative_to(res)): f.read_bytes() for f in sorted(res.rglob("*"))
                     if f.is_file() and f.name not in ("input.bin.pkl", "input.seqboarder.bin.pkl", "config.toml")}
    assert sorted(outs["single"]) == sorted(outs["dist"]) and {"kmer_count/k13.pkl", "kmer_count/k14.pkl"} <= set(outs["single"])
    for name, blob in outs["single"].items():
        if name.endswith(".pkl"):
            a, b = pickle.loads(blob), pickle.loads(outs["dist"][name])
            for x, y in zip(a, b):
                np.testing.assert_array_equal(np.asarray(x, dtype=object) if isinstance(x, list) else x,
                                              np.asarray(y, dtype=object) if isinstance(y, list) else y, err_msg=name)
        else:
            assert blob == outs["dist"][name], name


def _nccl_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import pickle
    import torch
    import torch.distributed as dist
    from kmap_amd import synth
    from kmap_amd.distributed import all_gather_concat, broadcast_seed, kmap_from_kmers_distributed, make_dist_device_seq
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        kh, cnts, lab, conseqs = _inputs()
        tr = {}
        best, _ = kmap_from_kmers_distributed(kh, cnts, lab, conseqs, K, n_max_iter=ITERS, random_seed=SEED, mode=1, trace=tr)
        out = {"best": best, "losses": tr["losses"], "seed": broadcast_seed(dist, None),
               "cat": all_gather_concat(dist, np.arange(7, dtype=np.uint16).reshape(-1, 1))}
        assert tr["collectives"] == 0                      # a one-rank group needs no collective ...
        tr2 = {}
        best2, _ = kmap_from_kmers_distributed(kh, cnts, lab, conseqs, K, n_max_iter=ITERS, random_seed=SEED, mode=1, trace=tr2,
                                               always_collective=True, profile_iters=4)
        assert tr2["collectives"] == ITERS                 # ... bench.py's overhead leg issues exactly ONE per iteration anyway
        out["best2"], out["phases"] = best2, tr2["phases"]
        seq, borders = synth.synth_reads(20_003, 75, 4)
        ds = make_dist_device_seq(seq, borders, dist)
        dc = DeviceCounts()
        ds.count(dc, 9, dedupe=True, merge_revcom=True)
        out["counts"] = dc.fetch()
        # bins owned by key range on RCCL: uint8 SUM all-reduce of the presence nibbles, in-place reduce of table slices that
        # live in library-owned memory, all_gather of the shards, adopt
        ds_r = make_dist_device_seq(seq, borders, dist, shard_counts=True)
        ds_r.count(dc, 12, dedupe=True, merge_revcom=True)
        out["counts_range"] = dc.fetch()
        ds_r.close()
        # ... and by key space on RCCL (the shard bookkeeping on device tensors; a one-rank group owns the whole key range)
        ds_k = make_dist_device_seq(seq, borders, dist, key_space=True)
        ds_k.count(dc, 12, dedupe=True, merge_revcom=True)
        out["counts_keyspace"] = dc.fetch()
        ds_k.close()
        out["scan"] = ds.scan(8, kmer2hash("ATCGATAG"), 2, True)
        # the device-gathered hit list (GatheredHits) through the background CSV writer, fetched on the writer's thread
        from kmap_amd.kmer_count import _pkg_file, init_motif_def_dict
        from kmap_amd.motif_discovery import ScanHits, gen_motif_occurence_file
        lazy = ds.scan_lazy(8, kmer2hash("ATCGATAG"), 2, True)
        assert isinstance(lazy, ScanHits) and lazy.unfetched and lazy.n_seq == 20_003
        out["lazy_summary"] = (lazy.n_reads_hit, lazy.total, lazy.max_hits)
        writers = []
        np.random.seed(3)
        gen_motif_occurence_file(["AATCGATAGC", "CCTACGTA"], init_motif_def_dict(_pkg_file("default_motif_def_table.csv")), None,
                                 Path(out_dir) / "occ_nccl.csv", True, dev_seq=ds, writers=writers)
        for w in writers:
            w.join()
        dc.close()
        ds.close()
        with open(Path(out_dir) / "nccl.pkl", "wb") as fh:
            pickle.dump(out, fh)
    finally:
        dist.destroy_process_group()


def test_rccl_backend_single_rank(tmp_path):
    """The collectives' DEVICE-tensor code paths (backend "nccl" = RCCL: all_gather_into_tensor of the neighbour table on the
    GPU, histogram all-reduce on library-owned memory, device-staged all_gather_concat, seed broadcast) with a one-rank process
    group on the test box's single GPU -- the multi-rank tests above run over gloo, which takes the host-tensor branches."""
    import pickle
    import torch.multiprocessing as mp
    import kmap_amd.visualization as V
    from kmap_amd import synth
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash
    from kmap_amd.motif_discovery import DeviceSeq
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    with open(tmp_path / "nccl.pkl", "rb") as fh:
        got = pickle.load(fh)
    kh, cnts, lab, conseqs = _inputs()
    tr = {}
    best, _ = V.kmap_from_kmers(kh, cnts, lab, conseqs, K, n_max_iter=ITERS, random_seed=SEED, mode=1, trace=tr)
    np.testing.assert_array_equal(got["best"], best)
    np.testing.assert_array_equal(got["losses"], tr["losses"])
    assert got["seed"] is None and got["cat"].dtype == np.uint16 and got["cat"].shape == (7, 1)   # one rank: the seed stays "default"
    seq, borders = synth.synth_reads(20_003, 75, 4)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    ds.count(dc, 9, dedupe=True, merge_revcom=True)
    u, c = dc.fetch()
    np.testing.assert_array_equal(got["counts"][0], u)
    np.testing.assert_array_equal(got["counts"][1], c)
    ds.count(dc, 12, dedupe=True, merge_revcom=True)
    u, c = dc.fetch()
    np.testing.assert_array_equal(got["counts_range"][0], u)
    np.testing.assert_array_equal(got["counts_range"][1], c)
    np.testing.assert_array_equal(got["counts_keyspace"][0], u)
    np.testing.assert_array_equal(got["counts_keyspace"][1], c)
    hits, pos = ds.scan(8, kmer2hash("ATCGATAG"), 2, True)
    np.testing.assert_array_equal(got["scan"][0], hits)
    np.testing.assert_array_equal(got["scan"][1], pos)
    assert got["lazy_summary"] == (int(np.count_nonzero(hits)), len(pos), int(hits.max()))
    np.testing.assert_array_equal(got["best2"], best)
    assert set(got["phases"]) == {"forces_ms", "collective_ms", "apply_ms", "iteration_ms"} and got["phases"]["collective_ms"] > 0
    from kmap_amd.kmer_count import _pkg_file, init_motif_def_dict
    from kmap_amd.motif_discovery import gen_motif_occurence_file
    np.random.seed(3)
    gen_motif_occurence_file(["AATCGATAGC", "CCTACGTA"], init_motif_def_dict(_pkg_file("default_motif_def_table.csv")), None,
                             tmp_path / "occ_single.csv", True, dev_seq=ds)
    assert (tmp_path / "occ_nccl.csv").read_text() == (tmp_path / "occ_single.csv").read_text()
    dc.close()
    ds.close()
