#!/bin/bash
# PMC passes on the per-read dedupe kernel alone (tools/probes/dedupe_only.py).  usage: bash tools/pmc_dedupe.sh [tag] [env assignments...]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-dd}
shift || true
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
declare -A SETS
SETS[sq_a]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SETS[sq_b]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
SETS[sq_c]="SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_ACTIVE_INST_MISC"
for s in sq_a sq_b sq_c; do
  rocprofv3 --pmc ${SETS[$s]} --output-format csv -d $OUT/$s -- python3 $R/tools/probes/dedupe_only.py ${DD_K:-8} 2 > $OUT/$s.txt 2> $OUT/$s.err || { tail -5 $OUT/$s.err; echo "pass $s failed"; }
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/probes/dedupe_only.py ${DD_K:-8} 3 > $OUT/trace.txt 2> $OUT/trace.err || true
find $OUT -type f ! -name '*counter_collection.csv' ! -name '*kernel_stats.csv' ! -name '*.txt' ! -name '*.err' -delete
python3 - <<PY
import csv, glob, collections
for s in ("sq_a","sq_b","sq_c"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % s, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "dedupe" not in kn and "hist_packed" not in kn: continue
            acc[kn[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(kn[:60], r["Counter_Name"])] += 1
        for kn, d in acc.items():
            print(s, kn, {c: round(v / cnt[(kn, c)]) for c, v in d.items()})
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dedupe" in r["Name"] or "hist_packed" in r["Name"]:
            print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
