#!/usr/bin/env python3
"""C5-sized scan twice in one process, KMAP_SCAN_PLANES=plain and =idx: which reads differ, and what the oracle says of them"""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from kmap_amd import _ffi, synth
from kmap_amd.kmer_count import kmer2hash
from oracle import oracle as O

n_reads, L, k, radius = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000, 300, 14, 5
motif = "AGGACCTACGTACA"
ds, raw = synth.synth_reads_dev(n_reads, L, 3, motifs=(motif, "AATCGATAGC"), keep_raw=True)
lib = _ffi.lib()
h = _ffi.vp()
_ffi.check(lib.kmap_scan_create(C.byref(h)))
ds.declare_layout(h.value)
cons = int(kmer2hash(motif))
res = {}
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    for mode in ("plain", "idx"):
        os.environ["KMAP_SCAN_PLANES"] = mode
        tot = _ffi.i64(0)
        _ffi.check(lib.kmap_scan_run_packed_dev(h.value, ds.codes.ptr, ds.inval_orig.ptr, ds.n, ds.borders.ptr, ds.n_seq, k, cons, radius, 1,
                                                C.byref(tot), ds.planes.ptr, None))
        hits = np.empty(n_reads, np.int32)
        pos = np.empty(tot.value, np.int32)
        _ffi.check(lib.kmap_scan_fetch(h.value, _ffi.ptr(hits), None, _ffi.ptr(pos)))
        print(rep, mode, "total", tot.value, "reads with hit", int(np.count_nonzero(hits)), flush=True)
        res[mode] = (hits, pos)
    a, b = res["plain"][0], res["idx"][0]
    diff = np.flatnonzero(a != b)
    print(rep, "reads that differ:", len(diff), diff[:20], flush=True)
    if len(diff):
        print("  read index mod 64:", np.bincount(diff % 64, minlength=64))
        offs = {m: np.concatenate([[0], np.cumsum(res[m][0], dtype=np.int64)]) for m in res}
        buf, md = np.empty(L, np.int32), C.c_int(0)
        for r in diff[:12]:
            read = raw(int(r) * (L + 1), int(r) * (L + 1) + L)
            m = O.lib().ko_scan_read(read, L, k, cons, radius, 1, buf, md)
            pa = res["plain"][1][offs["plain"][r]:offs["plain"][r + 1]]
            pb = res["idx"][1][offs["idx"][r]:offs["idx"][r + 1]]
            print("  read", int(r), "oracle", m, list(buf[:m]), "plain", list(pa), "idx", list(pb),
                  "abs pos of first idx-only:", [int(r) * L + int(x) for x in pb if x not in pa][:3])
