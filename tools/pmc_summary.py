#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_round.sh into one table per workload:
    python tools/pmc_summary.py gpurun_out/pmc_r02 profiles/r02_pmc.json
Per kernel: launches, mean of every counter per launch, and derived ratios
  valu_busy      = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (share of resident wave-cycles spent issuing VALU; both count quad-cycles)
  valu_issue_util= SQ_ACTIVE_INST_VALU * 4 / (n_simd * SQ_BUSY_CYCLES / 32)   (share of the chip's SIMD issue slots that issued VALU:
                   the SQ counters tick in quad-cycles and are summed over 32 shader engines; n_simd = 1024 on MI355X) -- the
                   "VALU utilisation" of DESIGN.md; a different quantity from valu_busy (3 waves per SIMD at 90 % utilisation have
                   valu_busy 0.3)
  resident_waves = SQ_WAVE_CYCLES * 4 / (SQ_BUSY_CYCLES / 32)   (average waves on the chip while it is busy; 4096 slots)
  valu_per_wave  = SQ_INSTS_VALU / SQ_WAVES               (VALU instructions per wave; x 64 / pairs-per-wave = lane instructions per pair)
  wait_any       = SQ_WAIT_ANY / SQ_WAVE_CYCLES           (waves parked on s_waitcnt / barriers)
  issue_stall    = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  hbm_bytes      = 2 * FETCH_SIZE + WRITE_SIZE  (KB -> B; gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)
"""
import collections
import csv
import glob
import json
import re
import sys
from pathlib import Path


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", name)
    base = m.group(1) if m else name
    targs = (m.group(2) or "") if m else ""
    targs = re.sub(r"\(anonymous namespace\)::", "", targs)
    return (base + targs)[:80]


def load(dirpath):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(str(Path(dirpath) / "*" / "*counter_collection.csv")):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])      # sum over the per-XCC / per-SE rows of one dispatch
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for (d, c), v in per_dispatch.items():
            agg[names[d]][c].append(v)
    return agg


def main():
    root = Path(sys.argv[1])
    out_path = Path(sys.argv[2]) if len(sys.argv) > 2 else None
    result = {"source": str(root), "units": "counter means per launch; FETCH_SIZE / WRITE_SIZE in KB as reported; hbm_bytes corrected",
              "workloads": {}}
    for wl in ("e2e", "seq", "seqshard", "count14", "keyspace14"):
        kern = collections.defaultdict(dict)
        for s in ("sq_a", "sq_b", "fetch", "write"):
            for k, cs in load(root / f"{wl}_{s}").items():
                for c, v in cs.items():
                    kern[k][c] = sum(v) / len(v)
                    kern[k]["launches"] = len(v)
        for k, d in kern.items():
            wc = d.get("SQ_WAVE_CYCLES")
            if wc:
                for name, c in (("valu_busy", "SQ_ACTIVE_INST_VALU"), ("wait_any", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"),
                                ("active_any", "SQ_ACTIVE_INST_ANY")):
                    if c in d:
                        d[name] = d[c] / wc
            if d.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in d:
                busy = d["SQ_BUSY_CYCLES"] / 32.0
                d["valu_issue_util"] = d["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * busy)
                if wc:
                    d["resident_waves"] = wc * 4.0 / busy
            if d.get("SQ_WAVES") and "SQ_INSTS_VALU" in d:
                d["valu_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
            if d.get("SQ_LDS_IDX_ACTIVE"):
                d["lds_conflict"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
            if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
                d["hbm_bytes"] = (2 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024
        result["workloads"][wl] = dict(sorted(kern.items()))
    if out_path:
        out_path.write_text(json.dumps(result, indent=1))
    for wl, kern in result["workloads"].items():
        print(f"== {wl}")
        print(f"{'kernel':60s} {'n':>5s} {'valu_busy':>9s} {'issue_util':>10s} {'res.waves':>9s} {'wait':>6s} {'stall':>6s} {'ldsconf':>7s} {'hbm MB':>10s} "
              f"{'VALU inst':>12s} {'VALU/wave':>10s} {'LDS inst':>12s}")
        nan = float("nan")
        for k, d in kern.items():
            print(f"{k[:60]:60s} {d.get('launches', 0):5d} {d.get('valu_busy', nan):9.2f} {d.get('valu_issue_util', nan):10.2f} {d.get('resident_waves', nan):9.0f} "
                  f"{d.get('wait_any', nan):6.2f} {d.get('issue_stall', nan):6.2f} {d.get('lds_conflict', nan):7.3f} {d.get('hbm_bytes', nan) / 1e6:10.1f} "
                  f"{d.get('SQ_INSTS_VALU', nan):12.3g} {d.get('valu_per_wave', nan):10.0f} {d.get('SQ_INSTS_LDS', nan):12.3g}")


if __name__ == "__main__":
    main()
