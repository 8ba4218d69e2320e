#!/usr/bin/env python3
"""Only SEQ force evaluations of a row-sharded session (default: rank 0 of 8 at C3, 6 250 x 50 000, producer / adder form): workload
for kernel-time / PMC passes.  `seq_shard_only.py [adder|classic] [n row0 nrows]`"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))   # seqa_check.forces
from kmap_amd import _ffi, visualization as V
import seqa_check as sc
form = sys.argv[1] if len(sys.argv) > 1 else "adder"
n, row0, nrows = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (50000, 0, 6250)))
lut = V.hd_prob_lut(8, 20, 3200)
rng = np.random.default_rng(1)
lds = (n + 127) & ~127
blk = rng.integers(0, 3201, size=(1024, lds), dtype=np.uint16)
sums = np.concatenate([np.roll(blk, 17 * i, axis=1) for i in range((nrows + 1023) // 1024)])[:nrows]
sums_d = _ffi.DeviceBuffer.from_numpy(sums)
scale = float(os.environ.get("SEQ_SCALE", "5"))          # 5: no far step; 1e5: every step far
ld = (rng.standard_normal((2, n)) * scale).astype(np.float32)
_, t = sc.forces(form, n, row0, nrows, sums_d, lds, lut, ld, 5)
print(form, n, nrows, "ms", round(t, 4))
