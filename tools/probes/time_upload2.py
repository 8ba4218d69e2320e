#!/usr/bin/env python3
"""Is the slow first H2D copy a one-time cost of the runtime's copy path?  usage: time_upload2.py <warm-up bytes> [mb of main copy]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kmap_amd import _ffi  # noqa: E402

warm = int(sys.argv[1])
n = int(sys.argv[2]) << 20 if len(sys.argv) > 2 else 1510 << 20
a = np.random.default_rng(0).integers(0, 4, size=n, dtype=np.uint8)
t0 = time.perf_counter()
_ffi.DeviceBuffer(1 << 20).free()
t1 = time.perf_counter()
if warm:
    w = _ffi.DeviceBuffer.from_numpy(a[:warm])
    _ffi.sync()
t2 = time.perf_counter()
for rep in range(3):
    t = time.perf_counter()
    raw = _ffi.DeviceBuffer.from_numpy(a)
    _ffi.sync()
    dt = time.perf_counter() - t
    print(f"warm-up {warm} B: runtime init {t1 - t0:.3f} s, warm-up copy {t2 - t1:.3f} s, copy {rep}: {dt:.3f} s = {n / dt / 1e9:.1f} GB/s")
    raw.free()
