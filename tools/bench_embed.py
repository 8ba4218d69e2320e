#!/usr/bin/env python3
"""Micro-benchmark of the embedding iteration at a given N (both modes), for rocprofv3 runs."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--modes", default="fast,seq")
    args = ap.parse_args()
    from kmap_amd import _ffi, visualization as V
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    n, k = args.n, args.k
    rng = np.random.default_rng(2)
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64).astype(np.uint32)
    lab = np.zeros(n, np.int32)
    ldd = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    D_d = _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, k, [k], D_d.ptr, ldd)
    t0 = time.perf_counter()
    nb_d = V.knn_select_dev(D_d.ptr, ldd, n, 20)
    _ffi.sync()
    t1 = time.perf_counter()
    sums_d, lds = V.knn_sums_dev(D_d.ptr, ldd, nb_d, n, 20)
    _ffi.sync()
    t2 = time.perf_counter()
    res = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, k, [k], nb_d, 20)          # warm-up (scratch allocation)
    t3 = time.perf_counter()
    res2 = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, k, [k], nb_d, 20) if res is not None else None
    _ffi.sync()
    t4 = time.perf_counter()
    same = res is not None and np.array_equal(res[0].to_numpy(np.uint16, (n, res[1]))[:64, :n], sums_d.to_numpy(np.uint16, (n, lds))[:64, :n])
    for r in (res, res2):
        if r is not None:
            r[0].free()
    print(f"N={n}: knn_select {1e3 * (t1 - t0):.2f} ms, knn_sums from the matrix {1e3 * (t2 - t1):.2f} ms, "
          f"from the k-mers {1e3 * (t4 - t3):.2f} ms (incl. 2N^2-byte allocation; first rows equal: {same})")
    lut = V.hd_prob_lut(k, 20, 400 * k)
    ld, ph = V._init_draws(n, 10, 7)
    for mode in args.modes.split(","):
        m = V.EMBED_FAST if mode == "fast" else V.EMBED_SEQ
        sess = V.EmbedSession(n, 10, 0.01, m)
        sess._keep = []
        check = _ffi.check
        check(_ffi.lib().kmap_embed_set_prob_lut(sess._h, sums_d.ptr, lds, _ffi.ptr(lut), len(lut)))
        sess.set_coords(ld, ph)
        sess.set_jitter(np.random.normal(0, 0.01, 4096))
        sess.step(2)
        _ffi.sync()
        t0 = time.perf_counter()
        sess.step(args.iters)
        _ffi.sync()
        dt = (time.perf_counter() - t0) / args.iters
        print(f"  mode={mode}: {1e3 * dt:.3f} ms/iter  ({n * n / dt / 1e9:.1f} Gpairs/s)  state={sess.state()}")
        sess.close()


if __name__ == "__main__":
    main()
