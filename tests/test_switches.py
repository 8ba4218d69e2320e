"""Every KMAP_* environment switch the product reads selects something a test runs (round-3 verdict: ~25 A/B switches selected
kernels no test ran; the superseded kernels are gone, this file covers the switches that stayed and are not exercised elsewhere:
KMAP_EMBED_MODE, KMAP_KNN, KMAP_IO_THREADS; KMAP_EMBED_SYM / _GRAPH / KMAP_SEQ_TAIL / _PAIR live in test_gpu_embed.py and
test_gpu_fullsize.py, KMAP_DIST_* in test_gpu_distributed.py / test_host_logic.py)."""
import ctypes as C
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_every_switch_of_the_product_is_named_in_a_test():
    """grep of kmap_amd/ for KMAP_* environment reads must be a subset of the names the test suite sets"""
    pat = re.compile(r'(?:getenv\("|environ\.get\("|environ\[")(KMAP_[A-Z0-9_]+)')
    used = set()
    for f in list((ROOT / "kmap_amd").rglob("*.py")) + list((ROOT / "kmap_amd" / "csrc").glob("*.hip")) + list((ROOT / "kmap_amd" / "csrc").glob("*.h")):
        used |= set(pat.findall(f.read_text()))
    named = set()
    for f in (ROOT / "tests").glob("test_*.py"):
        named |= set(re.findall(r"KMAP_[A-Z0-9_]+", f.read_text()))
    assert used, "no switch found: the pattern is stale"
    assert used <= named, f"switches without a test: {sorted(used - named)}"


def test_embed_mode_switch(monkeypatch):
    """SEQ -- the reference's arithmetic (visualization.py:296-317) -- is the default at EVERY N; FAST is opt-in through config.toml's
    optional visualization.embed_mode key or KMAP_EMBED_MODE (the environment wins)"""
    from kmap_amd import _policy, visualization as V
    monkeypatch.delenv("KMAP_EMBED_MODE", raising=False)
    _policy.reset()
    assert V.default_mode(50_000) == V.EMBED_SEQ and V.default_mode(200_000) == V.EMBED_SEQ and V.default_mode() == V.EMBED_SEQ
    monkeypatch.setenv("KMAP_EMBED_MODE", "seq")
    assert V.default_mode(10 ** 6) == V.EMBED_SEQ
    monkeypatch.setenv("KMAP_EMBED_MODE", "FAST")
    assert V.default_mode(10) == V.EMBED_FAST
    monkeypatch.setenv("KMAP_EMBED_MODE", "bogus")
    assert V.default_mode(10) == V.EMBED_SEQ
    monkeypatch.delenv("KMAP_EMBED_MODE")
    try:
        _policy.apply_config({"visualization": {"embed_mode": "fast"}})
        assert V.default_mode(10) == V.EMBED_FAST
        monkeypatch.setenv("KMAP_EMBED_MODE", "seq")
        assert V.default_mode(10) == V.EMBED_SEQ
        monkeypatch.delenv("KMAP_EMBED_MODE")
        _policy.apply_config({"general": {}, "visualization": {}})          # keys absent (the reference's config.toml)
        assert V.default_mode(10 ** 6) == V.EMBED_SEQ
        with pytest.raises(ValueError):
            _policy.apply_config({"visualization": {"embed_mode": "quick"}})
    finally:
        _policy.reset()


def test_exact_switch(monkeypatch):
    """KMAP_EXACT=1 / config general.exact: the reference's numpy calls at every size -- np.argpartition neighbours
    (visualization.py:100), np.argpartition top-k (motif_discovery.py:661), np.random.multinomial (:912); the GPU tests
    test_find_motif_device_topk_path / test_sample_disp_kmer_vs_oracle / test_exact_run_above_the_thresholds run the paths"""
    from kmap_amd import _policy, motif_discovery as MD, visualization as V
    monkeypatch.delenv("KMAP_EXACT", raising=False)
    monkeypatch.delenv("KMAP_KNN", raising=False)
    _policy.reset()
    big = V.KNN_NUMPY_MAX_N + 1
    assert not _policy.exact() and V.knn_mode(big) == "device" and MD._device_topk(MD.TOPK_DEVICE_MIN + 1, 5)
    monkeypatch.setenv("KMAP_EXACT", "1")
    assert _policy.exact() and V.knn_mode(big) == "numpy" and not MD._device_topk(MD.TOPK_DEVICE_MIN + 1, 5)
    monkeypatch.setenv("KMAP_KNN", "device")                       # the specific switch still wins over the general one
    assert V.knn_mode(big) == "device"
    monkeypatch.delenv("KMAP_KNN")
    monkeypatch.delenv("KMAP_EXACT")
    try:
        _policy.apply_config({"general": {"exact": True}})
        assert _policy.exact() and V.knn_mode(big) == "numpy"
        monkeypatch.setenv("KMAP_EXACT", "0")
        assert not _policy.exact()
    finally:
        _policy.reset()


def _csv_case(n_seq=70_000, n_cons=2, seed=3):
    rng = np.random.default_rng(seed)
    hits = [np.where(rng.random(n_seq) < 0.3, rng.integers(1, 4, n_seq), 0).astype(np.int32) for _ in range(n_cons)]
    pos = [np.sort(rng.integers(0, 140, int(h.sum()))).astype(np.int32) for h in hits]
    read_len = rng.integers(100, 151, n_seq).astype(np.int64)
    return hits, pos, read_len


_CSV_CHILD = """
import ctypes as C, sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.test_switches import _csv_case
from kmap_amd import _ffi
hits, pos, read_len = _csv_case()
hp = (C.c_void_p * 2)(*[h.ctypes.data for h in hits]); pp = (C.c_void_p * 2)(*[p.ctypes.data for p in pos])
rows = _ffi.i64(0)
_ffi.check(_ffi.lib().kmap_write_occurrence_csv(sys.argv[2].encode(), b"seq_ind;motif_0_ACGT;motif_1_TTGA;seq_len", len(read_len), 2, hp, pp,
                                                 _ffi.ptr(read_len), C.byref(rows)))
print(rows.value)
"""


def test_io_threads_switch(tmp_path):
    """the native occurrence-CSV writer (pure host code: runs without a GPU) with KMAP_IO_THREADS=1 and with its default pool writes
    the same bytes, and the rows are the reference's format (motif_discovery.py:1396-1419): `seq_ind;loc,loc;...;seq_len`, only
    reads with at least one hit"""
    outs = {}
    for tag, env in (("one", {"KMAP_IO_THREADS": "1"}), ("pool", {"KMAP_IO_THREADS": "7"}), ("default", {})):
        path = tmp_path / f"{tag}.csv"
        e = dict(os.environ, **env)
        if not env:
            e.pop("KMAP_IO_THREADS", None)
        r = subprocess.run([sys.executable, "-c", _CSV_CHILD, str(ROOT), str(path)], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = (path.read_bytes(), int(r.stdout.strip().splitlines()[-1]))
    assert outs["one"] == outs["pool"] == outs["default"]
    hits, pos, read_len = _csv_case()
    lines = outs["one"][0].decode().splitlines()
    any_hit = (hits[0] + hits[1]) > 0
    assert lines[0] == "seq_ind;motif_0_ACGT;motif_1_TTGA;seq_len" and len(lines) == 1 + int(any_hit.sum()) == 1 + outs["one"][1]
    offs = [np.concatenate([[0], np.cumsum(h, dtype=np.int64)]) for h in hits]
    for ln in (lines[1], lines[len(lines) // 2], lines[-1]):
        f = ln.split(";")
        s = int(f[0])
        assert any_hit[s] and int(f[3]) == read_len[s]
        for c in range(2):
            want = ",".join(str(int(x)) for x in pos[c][offs[c][s]:offs[c][s + 1]])
            assert f[1 + c] == want


@pytest.mark.gpu
def test_knn_numpy_mode_above_the_dense_limit(monkeypatch):
    """the neighbours above the dense hand-over limit (N > 16 384; numpy is the default up to N = 65 536, KMAP_KNN=numpy beyond) are
    the reference's np.argpartition on int64 rows (visualization.py:100), taken from the device matrix streamed back in row blocks.  Checked on sampled rows against argpartition of the ORACLE's rows."""
    from kmap_amd import visualization as V
    from kmap_amd.motif_discovery import DENSE_PKL_MAX_N
    from oracle import oracle as O
    n, k = DENSE_PKL_MAX_N + 616, 8
    rng = np.random.default_rng(21)
    kh = np.sort(rng.integers(0, 4 ** k, size=n).astype(np.uint32))
    lab = np.sort(rng.integers(0, 3, size=n)).astype(np.int64)
    conseqs = ["ACGTACGT", "ACGTAC", "TTGACCA"]
    monkeypatch.setenv("KMAP_KNN", "numpy")
    assert V.knn_mode(n) == "numpy"
    tr = {}
    V.kmap_from_kmers(kh, np.ones(n, np.int64), lab, conseqs, k, n_max_iter=2, random_seed=5, mode=V.EMBED_FAST, trace=tr)
    nb = tr["nb"]
    assert nb.shape == (n, 20)
    rows = np.unique(np.concatenate([[0, n - 1], rng.integers(0, n, 60)]))
    want = np.empty((1, n), np.uint8)
    cl = np.array([len(c) for c in conseqs], np.int32)
    for r in rows:
        O.lib().ko_hamdist_rows(np.ascontiguousarray(kh, np.uint64), lab.astype(np.int32), n, k, cl, len(cl), int(r), 1, want)
        np.testing.assert_array_equal(nb[r], np.argpartition(want[0].astype(np.int64), 20)[:20])
    monkeypatch.setenv("KMAP_KNN", "device")
    assert V.knn_mode(10) == "device"
    monkeypatch.delenv("KMAP_KNN")
    assert V.knn_mode(n) == "numpy" and V.knn_mode(V.KNN_NUMPY_MAX_N) == "numpy" and V.knn_mode(V.KNN_NUMPY_MAX_N + 1) == "device"
    # the default at this N (numpy up to N = 65 536) and KMAP_EXACT=1 select the same neighbours (threaded row blocks == one
    # argpartition call per row)
    monkeypatch.setenv("KMAP_EXACT", "1")
    assert V.knn_mode(n) == "numpy" and V.knn_mode(V.KNN_NUMPY_MAX_N + 1) == "numpy"
    tr2 = {}
    V.kmap_from_kmers(kh, np.ones(n, np.int64), lab, conseqs, k, n_max_iter=2, random_seed=5, mode=V.EMBED_FAST, trace=tr2)
    np.testing.assert_array_equal(tr2["nb"], nb)


@pytest.mark.gpu
@pytest.mark.parametrize("n,reps", [(1000, "counts"), (1003, "counts"), (777, "none"), (64, "all")])
def test_knn_numpy_repeated_rows(n, reps):
    """a sample repeats its k-mers count times (reference motif_discovery.py:759-772) and the row of a repeated k-mer equals the row
    above it: knn_select_numpy partitions the distinct rows only (kmap_rows_fresh_u8_dev / kmap_gather_rows_u8_dev) and must return
    exactly np.argpartition of every row -- with runs of repeats, without any, with one row repeated throughout, n % 16 != 0."""
    from kmap_amd import _ffi, visualization as V
    rng = np.random.default_rng(n)
    if reps == "counts":
        base = rng.integers(0, 9, size=(n // 3, n), dtype=np.uint8)
        cnt = rng.integers(1, 6, size=len(base))
        rows = np.repeat(base, cnt, axis=0)[:n]
        rows = np.concatenate([rows, rng.integers(0, 9, size=(n - len(rows), n), dtype=np.uint8)])
        rows[n // 2] = rows[n // 2 - 1]
        rows[n // 2, n - 1] ^= 1                                   # differs from the row above in its last byte only
    elif reps == "none":
        rows = rng.integers(0, 9, size=(n, n), dtype=np.uint8)
    else:
        rows = np.repeat(rng.integers(0, 9, size=(1, n), dtype=np.uint8), n, axis=0)
    ldd = (n + 127) & ~127
    D = np.zeros((n, ldd), np.uint8)
    D[:, :n] = rows
    D_d = _ffi.DeviceBuffer.from_numpy(D)
    fresh_d = _ffi.DeviceBuffer(n)
    try:
        _ffi.check(_ffi.lib().kmap_rows_fresh_u8_dev(D_d.ptr, ldd, n, 0, n, fresh_d.ptr, None))
        fresh = fresh_d.to_numpy(np.uint8, (n,))
        want_fresh = np.concatenate([[1], (rows[1:] != rows[:-1]).any(axis=1)]).astype(np.uint8)
        np.testing.assert_array_equal(fresh, want_fresh)
        for nrows in (n, n // 2 + 1):
            nb = V.knn_select_numpy(D_d.ptr, ldd, n, 20, nrows)
            np.testing.assert_array_equal(nb, np.argpartition(rows[:nrows].astype(np.int64), 20, axis=1)[:, :20])
    finally:
        D_d.free()
        fresh_d.free()
