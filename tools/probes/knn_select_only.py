#!/usr/bin/env python3
"""Only the neighbour selection at N = 50 000 on a pipeline-like sample (k-mers grouped by label, expanded by counts: many zero-distance
ties) and on unique sorted k-mers: ms per launch of the kernel KMAP_KNN_SELECT picks (1 one pass, 2 two passes; default by row length),
checked against a stable argsort on sampled rows.  `knn_select_only.py [n]`"""
import os
import statistics
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kmap_amd import _ffi   # noqa: E402
from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50000
rng = np.random.default_rng(3)
cases = {}
if "--c3" in sys.argv:                              # the real hand-over sample of the C3 pipeline (what bench.py's knn_select stage times)
    import pickle
    import shutil
    from kmap_amd.e2e import run_e2e
    r = run_e2e("C3", "fast", iters=1, keep=True)
    with open(Path(r["res_dir"]) / "sample_kmers.pkl", "rb") as fh:
        skh, scnt, slab, sconseq = pickle.load(fh)
    shutil.rmtree(r["res_dir"], ignore_errors=True)
    cases["C3 hand-over"] = (np.repeat(np.asarray(skh), scnt).astype(np.uint32), np.repeat(np.asarray(slab), scnt).astype(np.int32), [len(c) for c in sconseq])
    n = len(cases["C3 hand-over"][0])
uniq = np.sort(rng.choice(4 ** 8, size=min(n, 32896), replace=False).astype(np.uint32))
w = rng.random(len(uniq)) ** 8                      # a few k-mers carry most of the weight, like motif k-mers
cnt = rng.multinomial(n, w / w.sum())
cases["pipeline-like"] = (np.repeat(uniq, cnt), np.sort(rng.integers(0, 3, size=n)).astype(np.int32), [8, 7, 8])
cases["unique sorted"] = (np.sort(rng.integers(0, 4 ** 8, size=n).astype(np.uint32)), np.zeros(n, np.int32), [8])
lib = _ffi.lib()
for name, (kh, lab, lens) in cases.items():
    ldd = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(np.ascontiguousarray(kh)), _ffi.DeviceBuffer.from_numpy(lab)
    D_d = _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, 8, lens, D_d.ptr, ldd)
    nb_d = _ffi.DeviceBuffer(n * 20 * 4)

    def sel():
        _ffi.check(lib.kmap_knn_select_u8_dev(D_d.ptr, ldd, n, 20, 0, n, nb_d.ptr, None))
    for _ in range(20):
        sel()
    evs = [_ffi.Event() for _ in range(11)]
    evs[0].record()
    for i in range(10):
        sel()
        evs[i + 1].record()
    _ffi.sync()
    ms = statistics.median(evs[i].elapsed_ms(evs[i + 1]) for i in range(10))
    got = np.sort(nb_d.to_numpy(np.int32, (n, 20)), axis=1)
    bad = 0
    for r in rng.integers(0, n, 200):
        drow = D_d.to_numpy(np.uint8, (n,), offset=int(r) * ldd)
        bad += not np.array_equal(got[r], np.sort(np.argsort(drow, kind="stable")[:20]))
    print(f"KMAP_KNN_SELECT={os.environ.get('KMAP_KNN_SELECT', 'default')} {name}: n={n} {ms:.4f} ms, {n * n / ms / 1e6:.0f} GB/s of N^2, rows differing from the stable argsort: {bad}", flush=True)
    for b in (kh_d, lab_d, D_d, nb_d):
        b.free()
