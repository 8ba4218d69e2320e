#!/usr/bin/env python3
"""Wall-time split of one occurrence scan at C3 scale: device pass, result fetch, >20 subsample logic, CSV writer."""
import ctypes as C
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kmap_amd import _ffi, synth  # noqa: E402
from kmap_amd.kmer_count import kmer2hash, init_motif_def_dict, _pkg_file  # noqa: E402
from kmap_amd import motif_discovery as md  # noqa: E402

seq, borders = synth.synth_reads(10_000_000, 150, 2)
ds = md.DeviceSeq(seq, borders)
mdd = init_motif_def_dict(_pkg_file("default_motif_def_table.csv"))
cons = ["AATCGATA", "CCTACGTA"]
for rep in range(3):
    t = [time.perf_counter()]
    tot = _ffi.i64(0)
    ds.scan(8, kmer2hash(cons[0]), 1, True) if rep == 0 else None
    t[0] = time.perf_counter()
    _ffi.check(_ffi.lib().kmap_scan_run_packed_dev(ds._scan, ds.codes.ptr, ds.inval_orig.ptr, ds.n, ds.borders.ptr, ds.n_seq, 8,
                                                   int(kmer2hash(cons[0])), 1, 1, C.byref(tot), ds.planes.ptr, None))
    _ffi.sync(); t.append(time.perf_counter())
    hits, pos = ds.scan(8, kmer2hash(cons[0]), 1, True); t.append(time.perf_counter())
    per = md.scan_motif_occurence(ds, cons, mdd, True); t.append(time.perf_counter())
    with tempfile.TemporaryDirectory() as d:
        md.gen_motif_occurence_file(cons, mdd, None, Path(d) / "o.csv", True, dev_seq=ds); t.append(time.perf_counter())
        size = (Path(d) / "o.csv").stat().st_size
    print(f"device {1e3*(t[1]-t[0]):.1f} ms | scan+fetch {1e3*(t[2]-t[1]):.1f} ms | 2 consensus scan+subsample {1e3*(t[3]-t[2]):.1f} ms | "
          f"same + csv ({size/1e6:.0f} MB) {1e3*(t[4]-t[3]):.1f} ms | hits {int(hits.sum())}")
