#!/usr/bin/env python3
"""Copy the judged summaries of gpurun_out/round_prof into profiles/ (tracked): python tools/collect_profiles.py r01"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = ROOT / "gpurun_out" / "round_prof"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)
for name in ("bench_trace", "e2e_trace", "embed_trace", "seqshard_trace", "scan_trace", "count_trace", "keyspace_trace"):
    f = glob.glob(str(src / name / "*" / "*kernel_stats.csv"))
    if f:
        shutil.copyfile(f[0], dst / f"{tag}_{name}_kernel_stats.csv")
for name in ("bench_trace.json", "e2e_trace.json", "embed_trace.txt", "seqshard_trace.txt", "scan_trace.json", "count_trace.txt", "keyspace_trace.txt"):
    if (src / name).exists():
        shutil.copyfile(src / name, dst / f"{tag}_{name}")
out = {"command": "rocprofv3 --pmc WRITE_SIZE (and, separately, FETCH_SIZE) --output-format csv -- python3 bench.py --no-cpu-baseline --no-c4 --no-stages --e2e none --steps 5 --warmup 1",
       "note": "units KB (x1024 B); FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B requests at 64 B); "
               "WRITE_SIZE exact for 16-B-per-lane stores", "kernels": {}}
for t in ("write", "fetch"):
    f = glob.glob(str(src / f"bench_pmc_{t}" / "*" / "*counter_collection.csv"))
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = k.split("<")[0].split("(")[0].strip()
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out["kernels"].setdefault(k, {})[c + "_KB_mean"] = sum(v) / len(v)
        out["kernels"][k][c + "_n"] = len(v)
hk = [k for k in out["kernels"] if "hamdist_tile_kernel" in k or "hamdist_matrix_kernel" in k]
if hk:
    d = out["kernels"][hk[0]]
    d["hbm_bytes_per_launch"] = (2 * d.get("FETCH_SIZE_KB_mean", 0) + d.get("WRITE_SIZE_KB_mean", 0)) * 1024
    d["algorithmic_bytes_per_launch"] = 50000 * 50000 + 5 * 50000
json.dump(out, open(dst / f"{tag}_bench_pmc.json", "w"), indent=1)
for name, to in (("pmc_summary.json", f"{tag}_pmc.json"), ("pmc_summary.txt", f"{tag}_pmc.txt")):
    if (src / name).exists():
        shutil.copyfile(src / name, dst / to)
print(sorted(p.name for p in dst.iterdir()))
