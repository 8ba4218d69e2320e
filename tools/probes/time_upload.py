#!/usr/bin/env python3
"""Where does DeviceSeq's upload of the C3 reads go?  (mapping with / without MAP_POPULATE, H2D copy, pack + planes, borders)"""
import pickle
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kmap_amd import _ffi, synth  # noqa: E402
from kmap_amd.kmer_count import load_array_pickle  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
seq, borders = synth.synth_reads(n_reads, 150, 2)
d = Path(tempfile.mkdtemp())
with open(d / "a.pkl", "wb") as fh:
    pickle.dump(seq, fh, protocol=4)
with open(d / "b.pkl", "wb") as fh:
    pickle.dump(borders, fh, protocol=4)
del seq, borders
_ffi.DeviceBuffer(1 << 20).free()          # runtime up
for populate in (True, False, True):
    t0 = time.perf_counter()
    a = load_array_pickle(d / "a.pkl", populate=populate)
    b = load_array_pickle(d / "b.pkl", populate=populate)
    t1 = time.perf_counter()
    raw = _ffi.DeviceBuffer.from_numpy(a)
    _ffi.sync()
    t2 = time.perf_counter()
    bd = _ffi.DeviceBuffer.from_numpy(b)
    _ffi.sync()
    t3 = time.perf_counter()
    n = len(a)
    groups = int(_ffi.lib().kmap_packed_groups(n))
    codes, inv, planes = _ffi.DeviceBuffer(groups * 4), _ffi.DeviceBuffer(groups * 2), _ffi.DeviceBuffer(groups * 4)
    _ffi.check(_ffi.lib().kmap_pack_reads_dev(raw.ptr, n, codes.ptr, inv.ptr, None))
    _ffi.check(_ffi.lib().kmap_pack_planes_dev(codes.ptr, n, planes.ptr, None))
    _ffi.sync()
    t4 = time.perf_counter()
    for x in (raw, bd, codes, inv, planes):
        x.free()
    t5 = time.perf_counter()
    print(f"populate={populate}: map {t1 - t0:.3f} s, H2D reads ({n / 1e9:.2f} GB) {t2 - t1:.3f} s = {n / (t2 - t1) / 1e9:.1f} GB/s, "
          f"H2D borders {t3 - t2:.3f} s, alloc + pack + planes {t4 - t3:.3f} s, free {t5 - t4:.3f} s")
    del a, b
