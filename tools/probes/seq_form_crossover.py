import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import numpy as np
from kmap_amd import _ffi, visualization as V
import seqa_check as sc
lut = V.hd_prob_lut(8, 20, 3200)
for n in (12000, 16384, 20000, 24000, 28000, 30000, 33000, 40000):
    rng = np.random.default_rng(n)
    lds = (n + 127) & ~127
    blk = rng.integers(0, 3201, size=(1024, lds), dtype=np.uint16)
    sums = np.concatenate([np.roll(blk, 17 * i, axis=1) for i in range((n + 1023) // 1024)])[:n]
    sums_d = _ffi.DeviceBuffer.from_numpy(sums)
    for scale in (5.0, 1e4):
        ld = (rng.standard_normal((2, n)) * scale).astype(np.float32)
        (gc, lc), tc = sc.forces("classic", n, 0, n, sums_d, lds, lut, ld, 5)
        (ga, la), ta = sc.forces("adder", n, 0, n, sums_d, lds, lut, ld, 5)
        print(f"n={n} scale={scale:g}: classic {tc:.3f} adder {ta:.3f} same={np.array_equal(gc.view(np.uint32), ga.view(np.uint32))}", flush=True)
    sums_d.free()
