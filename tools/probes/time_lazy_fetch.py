#!/usr/bin/env python3
"""Split of the background CSV job at C3: lazy scan, fetch of the lists (ScanHits.host), native writer."""
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
    t0 = time.perf_counter()
    per = md.scan_motif_occurence(ds, cons, mdd, True)
    t1 = time.perf_counter()
    host = [tuple(r) for r in per]
    t2 = time.perf_counter()
    with tempfile.TemporaryDirectory() as d:
        t3 = time.perf_counter()
        md.gen_motif_occurence_file(cons, mdd, None, Path(d) / "o.csv", True, dev_seq=ds)
        t4 = time.perf_counter()
    t5 = time.perf_counter()
    del host
    t6 = time.perf_counter()
    print(f"lazy scans {1e3*(t1-t0):.1f} | fetch {1e3*(t2-t1):.1f} | scan+fetch+csv {1e3*(t4-t3):.1f} | rmtree {1e3*(t5-t4):.1f} | free {1e3*(t6-t5):.1f} ms")
