#!/usr/bin/env python3
"""How many force-kernel steps of a converged C3 embedding are 'far' (every squared distance >= 1000, where q clips to 0.001 and the
SEQ kernels take their 12-instruction shortcut), by step shape: quad form (16 rows x 32 columns), pair form (32 x 64), producer step
(2 rows x 256 columns), a whole producer chunk (25 rows x 256 columns)?  Runs the C3 pipeline (FAST embedding: the statistics of the
layout, not its digits, matter here) and samples tiles of its low_dim_data.tsv."""
import shutil
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kmap_amd.e2e import run_e2e   # noqa: E402

r = run_e2e("C3", "fast", keep=True)
tab = np.loadtxt(Path(r["res_dir"]) / "low_dim_data.tsv", skiprows=1)
shutil.rmtree(r["res_dir"], ignore_errors=True)
x, y = tab[:, 0].astype(np.float32), tab[:, 1].astype(np.float32)
n = len(x)
print("N", n, "extent x", x.min(), x.max(), "y", y.min(), y.max())
rng = np.random.default_rng(1)
for name, (rr, cc) in {"quad 16x32": (16, 32), "pair 32x64": (32, 64), "producer step 2x256": (2, 256), "producer chunk 25x256": (25, 256),
                       "single pair": (1, 1)}.items():
    far = 0
    trials = 20000
    for _ in range(trials):
        i0 = int(rng.integers(0, max(1, n - rr)))
        j0 = int(rng.integers(0, max(1, n - cc)))
        dx = x[i0:i0 + rr, None] - x[None, j0:j0 + cc]
        dy = y[i0:i0 + rr, None] - y[None, j0:j0 + cc]
        far += bool(((dx * dx + dy * dy) >= 1000.0).all())
    print(f"{name}: {far / trials:.3f} of the sampled tiles are all-far")

# ---- ms per SEQ iteration along the run (segments of 250 iterations) and the extent of the layout at the end of each segment
import pickle   # noqa: E402
import time   # noqa: E402
from kmap_amd import _ffi, visualization as V   # noqa: E402
from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for   # noqa: E402
r = run_e2e("C3", "fast", iters=1, keep=True)
with open(Path(r["res_dir"]) / "sample_kmers.pkl", "rb") as fh:
    skh, scnt, slab, sconseq = pickle.load(fh)
shutil.rmtree(r["res_dir"], ignore_errors=True)
kh = np.repeat(np.asarray(skh), scnt).astype(np.uint32)
lab = np.repeat(np.asarray(slab), scnt).astype(np.int32)
lens = [len(c) for c in sconseq]
n = len(kh)
ldd = pitch_for(n)
kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
D_d = _ffi.DeviceBuffer(n * ldd)
hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, 8, lens, D_d.ptr, ldd)
nb = V.knn_select_dev(D_d.ptr, ldd, n, 20)
sums_d, lds = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, 8, lens, nb, 20)
lut = V.hd_prob_lut(8, 20, 3200)
ld0, ph = V._init_draws(n, 10, 7)
sess = V.EmbedSession(n, 10, 0.01, V.EMBED_SEQ)
sess.set_prob_lut(sums_d, lds, lut)
sess.set_coords(ld0, ph)
sess.set_jitter(np.random.normal(0, 0.01, 8192))
for seg in range(10):
    _ffi.sync()
    t0 = time.perf_counter()
    sess.step(250)
    _ffi.sync()
    dt = time.perf_counter() - t0
    c = sess.coords()
    print(f"iterations {seg * 250 + 1}..{(seg + 1) * 250}: {dt / 250 * 1e3:.3f} ms per iteration, extent {np.abs(c).max():.1f}, "
          f"far pairs (sample) {np.mean((np.subtract.outer(c[0, ::50], c[0, ::50]) ** 2 + np.subtract.outer(c[1, ::50], c[1, ::50]) ** 2) >= 1000):.3f}", flush=True)
