#!/usr/bin/env python3
"""Where and when the waves of ONE SEQ force evaluation (N = 50 000, classic forms, all batches far) ran: needs the debug build of the
library (per-wave start / end in 100-MHz ticks, HW_ID, XCC_ID in a device array; KMAP_PROBE_LIB=...).  Prints per form (wide / pair / quad
blocks) the spread of start and end times and how many waves of each form every SIMD carried."""
import os, sys, ctypes as C
from collections import Counter, defaultdict
from pathlib import Path
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from kmap_amd import _ffi, visualization as V
_ffi.LIB_PATH = Path(os.environ["KMAP_PROBE_LIB"])
import seqa_check as sc
n = 50000
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1e5
lut = V.hd_prob_lut(8, 20, 3200)
rng = np.random.default_rng(1)
lds = (n + 127) & ~127
blk = rng.integers(0, 3201, size=(1024, lds), dtype=np.uint16)
sums = np.concatenate([np.roll(blk, 17 * i, axis=1) for i in range((n + 1023) // 1024)])[:n]
sums_d = _ffi.DeviceBuffer.from_numpy(sums)
ld = (rng.standard_normal((2, n)) * scale).astype(np.float32)
_, t = sc.forces("classic", n, 0, n, sums_d, lds, lut, ld, 3)
print("ms", t)
L = C.CDLL(str(_ffi.LIB_PATH))
buf = np.zeros(8192 * 4, np.uint64)
L.kmap_debug_read.argtypes = [C.c_void_p, C.c_size_t]
rc = L.kmap_debug_read(buf.ctypes.data, buf.nbytes)
d = buf.reshape(-1, 4)
nb_tail, nb_pair, nb_quad = 212, 256, 256
t0 = d[:(nb_tail + nb_pair + nb_quad) * 4, 0].astype(np.int64)
base = t0[t0 > 0].min()
per_simd = defaultdict(Counter)
for form, lo, hi in (("wide", 0, nb_tail), ("pair", nb_tail, nb_tail + nb_pair), ("quad", nb_tail + nb_pair, nb_tail + nb_pair + nb_quad)):
    rows = d[lo * 4:hi * 4]
    st = (rows[:, 0].astype(np.int64) - base) / 100.0     # us
    en = (rows[:, 1].astype(np.int64) - base) / 100.0
    print(f"{form}: {len(rows)} waves; start us min {st.min():.1f} median {np.median(st):.1f} max {st.max():.1f}; end us min {en.min():.1f} "
          f"median {np.median(en):.1f} max {en.max():.1f}; duration median {np.median(en - st):.1f} max {(en - st).max():.1f}")
    for r in rows:
        hw, xcc = int(r[2]), int(r[3]) & 0xf
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
        per_simd[(xcc, se, sh, cu, simd)][form] += 1
mix = Counter(tuple(sorted(c.items())) for c in per_simd.values())
print("SIMDs seen:", len(per_simd))
for m, k in sorted(mix.items(), key=lambda kv: -kv[1]):
    print(f"  {k:5d} SIMDs carried {dict(m)}")
