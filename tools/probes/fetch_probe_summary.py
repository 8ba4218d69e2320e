#!/usr/bin/env python3
"""FETCH_SIZE (KB, as rocprofv3 reports it) of tools/probes/fetch_probe's launches against the 2 GiB each of them reads:
    python tools/probes/fetch_probe_summary.py <rocprofv3 output dir>"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            agg[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
known_kb = 2 * 1024 * 1024
print(f"{'kernel':24s} {'FETCH_SIZE KB / launch':>24s} {'x 1024 / bytes read':>22s}   (bytes the kernel's loads ask for: 2 GiB; *_touch: 1/8, 1/32 of that)")
for k, d in sorted(agg.items()):
    v = sum(d.values()) / len(d)
    print(f"{k:24s} {v:24.0f} {v / known_kb:22.3f}")
