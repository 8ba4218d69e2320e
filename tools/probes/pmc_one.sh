#!/bin/bash
# two PMC passes (never combined with a trace) over one probe command; prints the mean counters of the kernels matching $1
#   bash tools/probes/pmc_one.sh <kernel substring> <python script under tools/probes> [args ...]
set -e
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_one
rm -rf $OUT && mkdir -p $OUT
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
rocprofv3 --pmc $A --output-format csv -d $OUT/a -- python3 $R/tools/probes/"$@" > $OUT/a.txt 2> $OUT/a.err
rocprofv3 --pmc $B --output-format csv -d $OUT/b -- python3 $R/tools/probes/"$@" > $OUT/b.txt 2> $OUT/b.err
python3 - $OUT "$PAT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        acc["_ms"] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6; n["_ms"] += 1
for c, v in sorted(acc.items()): print(f"   {c:24s} {v / n[c]:.5g}")
w = acc["SQ_WAVES"] / max(n["SQ_WAVES"], 1)
if w: print("   VALU per wave", acc["SQ_INSTS_VALU"] / n["SQ_INSTS_VALU"] / w, " SALU per wave", acc["SQ_INSTS_SALU"] / n["SQ_INSTS_SALU"] / w, " LDS per wave", acc["SQ_INSTS_LDS"] / n["SQ_INSTS_LDS"] / w)
PY
find $OUT -type f -name '*.csv' -delete
