#!/usr/bin/env python3
"""bench.py's C5 leg alone (50 M x 300 bp, k = 14, r = 5, one consensus; checked against the oracle on 200 reads):
    KMAP_SCAN_PLANES=plain|idx python3 tools/probes/c5_only.py [reps]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench

d = bench.c5_leg(reps=int(sys.argv[1]) if len(sys.argv) > 1 else 5)
print(json.dumps({k: d[k] for k in ("ms_median", "ms_min", "frac", "total_hits", "reads_with_hit") if k in d}))
