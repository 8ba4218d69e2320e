#!/usr/bin/env python3
"""SEQ force evaluation at N = 50 000 on one GPU against the SPREAD of the layout: standard-normal coordinates times `scale`
(5: no far batch; 1e5: every batch far), and a clustered layout like the one a real run converges to.  Says what the far shortcut of
seq_terms_dispatch is worth in time.  `seq_far_whatif.py [classic|adder]`"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from kmap_amd import _ffi, visualization as V
import seqa_check as sc
if os.environ.get("KMAP_PROBE_LIB"):        # a what-if build of the library (probe only)
    from pathlib import Path
    _ffi.LIB_PATH = Path(os.environ["KMAP_PROBE_LIB"])
form = sys.argv[1] if len(sys.argv) > 1 else "classic"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
lut = V.hd_prob_lut(8, 20, 3200)
rng = np.random.default_rng(1)
lds = (n + 127) & ~127
blk = rng.integers(0, 3201, size=(1024, lds), dtype=np.uint16)
sums = np.concatenate([np.roll(blk, 17 * i, axis=1) for i in range((n + 1023) // 1024)])[:n]
sums_d = _ffi.DeviceBuffer.from_numpy(sums)
base = rng.standard_normal((2, n))
for scale in ((5, 1e5) if os.environ.get("KMAP_PROBE_LIB") else (5, 30, 100, 300, 1000, 1e5)):
    ld = (base * scale).astype(np.float32)
    s = ld[:, ::50]
    farp = np.mean((np.subtract.outer(s[0], s[0]) ** 2 + np.subtract.outer(s[1], s[1]) ** 2) >= 1000)
    _, t = sc.forces(form, n, 0, n, sums_d, lds, lut, ld, 5)
    print(f"{form} n {n} scale {scale:g}: far pairs {farp:.3f}, {t:.3f} ms", flush=True)
