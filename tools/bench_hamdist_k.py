#!/usr/bin/env python3
"""Hamming-matrix launch time vs k at N = 50 000 (one-hot tile kernel with 1 / 2 code words, general kernel for k > 16)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kmap_amd import _ffi  # noqa: E402
from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for  # noqa: E402
from kmap_amd.kmer_count import get_hash_dtype  # noqa: E402

n = 50_000
rng = np.random.default_rng(1)
ld = pitch_for(n)
out_d = _ffi.DeviceBuffer(n * ld)
for k, lens in ((6, [6, 6]), (8, [8, 8]), (8, [8, 6]), (12, [12, 12]), (15, [15, 9]), (16, [16, 16]), (20, [20, 20]), (31, [31, 31])):
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64).astype(get_hash_dtype(k))
    lab = np.sort(rng.integers(0, 3, size=n)).astype(np.int32)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    for _ in range(5):
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, k, lens, out_d.ptr, ld)
    e0, e1 = _ffi.Event(), _ffi.Event()
    e0.record()
    for _ in range(50):
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, k, lens, out_d.ptr, ld)
    e1.record()
    _ffi.sync()
    ms = e0.elapsed_ms(e1) / 50
    print(f"k={k:2d} lens={lens}: {ms:.4f} ms  {n * n / ms / 1e6:.0f} GB/s  ({n * n / ms / 1e9:.2f}e12 pairs/s)")
