#!/usr/bin/env python3
"""How far do the FAST (symmetric, wavefront sums) and SEQ (reference summation order) embedding trajectories drift apart?
Same k-mers, same start, same jitter pool; max |coordinate difference| and relative loss difference after T iterations."""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--checkpoints", default="1,10,50,100,200,500")
    args = ap.parse_args()
    from kmap_amd import _ffi, visualization as V
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    n, k = args.n, args.k
    rng = np.random.default_rng(2)
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64).astype(np.uint32)
    kh[::5] = kh[rng.integers(0, n, size=len(kh[::5]))]                    # duplicates, as in a sampled k = 8 table
    lab = np.zeros(n, np.int32)
    ldd = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    D_d = _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, k, [k], D_d.ptr, ldd)
    nb_d = V.knn_select_dev(D_d.ptr, ldd, n, 20)
    sums_d, lds = V.knn_sums_dev(D_d.ptr, ldd, nb_d, n, 20)
    lut = V.hd_prob_lut(k, 20, 400 * k)
    ld, ph = V._init_draws(n, 10, 7)
    jit = np.random.default_rng(5).normal(0, 0.01, 4096)
    cps = [int(x) for x in args.checkpoints.split(",")]
    traj = {}
    for mode, m in (("seq", V.EMBED_SEQ), ("fast", V.EMBED_FAST)):
        sess = V.EmbedSession(n, 10, 0.01, m)
        _ffi.check(_ffi.lib().kmap_embed_set_prob_lut(sess._h, sums_d.ptr, lds, _ffi.ptr(lut), len(lut)))
        sess.set_coords(ld, ph)
        sess.set_jitter(jit)
        done, out = 0, []
        for cp in cps:
            sess.step(cp - done)
            done = cp
            out.append((sess.coords().copy(), sess.state()["last_loss"]))
        traj[mode] = out
        sess.close()
    scale = float(np.abs(traj["seq"][-1][0]).max())
    print(f"N = {n}, k = {k}; coordinate scale (max |y| at the end) {scale:.3f}")
    for cp, (ys, ls), (yf, lf) in zip(cps, traj["seq"], traj["fast"]):
        print(f"  after {cp:4d} iterations: max |y_fast - y_seq| = {np.abs(ys - yf).max():.3e}   loss rel diff = {abs(ls - lf) / abs(ls):.3e}")


if __name__ == "__main__":
    main()
