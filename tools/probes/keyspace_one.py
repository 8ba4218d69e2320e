#!/usr/bin/env python3
"""one key-space shard's count pass, a few times (for rocprofv3 --kernel-trace --stats):  keyspace_one.py K G RANK [full]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
k, G, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
from kmap_amd import _ffi
from kmap_amd.e2e import synth_config_reads
from kmap_amd.kmer_count import DeviceCounts
from kmap_amd.motif_discovery import DeviceSeq
seq, borders = synth_config_reads("C3")
ds, dc = DeviceSeq(seq, borders), DeviceCounts()
n_bins = 4 ** k
b = [(n_bins * i // G) & ~7 for i in range(G)] + [n_bins]
for _ in range(6):
    if len(sys.argv) > 4:
        ds.count(dc, k, dedupe=False, merge_revcom=True)
    else:
        ds.count_range(dc, k, False, True, b[r], b[r + 1] - b[r])
_ffi.sync()
print("n_uniq", dc.n_uniq)
