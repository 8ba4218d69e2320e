#!/bin/bash
# two SQ counter passes over any python script of the repo: bash tools/pmc_any.sh <tag> <kernel-pattern> <script.py> [args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}; PAT=$2; shift; shift
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
rocprofv3 --pmc $A --output-format csv -d $OUT/e2e_sq_a -- python3 $R/"$@" > $OUT/a.txt 2> $OUT/a.err
rocprofv3 --pmc $B --output-format csv -d $OUT/e2e_sq_b -- python3 $R/"$@" > $OUT/b.txt 2> $OUT/b.err
find $OUT -type f ! -name '*counter_collection.csv' ! -name '*.txt' -delete
python3 $R/tools/pmc_summary.py $OUT $OUT/summary.json | grep -E "kernel  |$PAT"
python3 - "$OUT/summary.json" "$PAT" <<'PY'
import json, sys, re
d = json.load(open(sys.argv[1]))
for wl, ks in d["workloads"].items():
    for k, v in ks.items():
        if re.search(sys.argv[2], k):
            print(k[:60], {c: (f"{x:.3g}" if isinstance(x, float) else x) for c, x in v.items() if c.startswith("SQ_")})
PY
