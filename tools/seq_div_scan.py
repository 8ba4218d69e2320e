#!/usr/bin/env python3
"""Which shortened division sequences of csrc/seq_div.h are exact on the SEQ kernel's operand ranges?  (exhaustive, on the GPU)"""
import ctypes as C
import struct
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kmap_amd import _ffi  # noqa: E402


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


L = _ffi.lib()
nb, fb = C.c_uint64(0), C.c_uint32(0)
print("1/s1 (clipped), s1 in [1, 2^100):")
for rs in range(4):
    _ffi.check(L.kmap_selftest_seq_div(0, rs, 0, bits(1.0), bits(2.0 ** 100), C.byref(nb), C.byref(fb)))
    print(f"  rcp steps {rs}: {nb.value} mismatches (first bits {fb.value:#x})")
print("q/(1-q), q in [0.001, 0.999]:")
for rs in range(3):
    for qs in range(4):
        _ffi.check(L.kmap_selftest_seq_div(1, rs, qs, bits(0.001), bits(0.999), C.byref(nb), C.byref(fb)))
        print(f"  rcp steps {rs}, quotient steps {qs} ({1 + 2 * rs + 1 + 2 * qs} instr): {nb.value} mismatches (first bits {fb.value:#x})")
