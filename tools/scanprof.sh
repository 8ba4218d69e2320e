#!/bin/bash
# kernel trace + two SQ counter passes of the occurrence scan at C3 (tools/bench_scan.py): bash tools/scanprof.sh [k] [radius] [read_len] [reads]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-8}; RAD=${2:-2}; LEN=${3:-150}; READS=${4:-10000000}
OUT=$R/gpurun_out/scanprof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/bench_scan.py --reads $READS --read_len $LEN --k $K --radius $RAD --reps 5 > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/t/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-56s %4s %9.1f us" % (r["Name"].split("::")[-1][:56], r["Calls"], float(r["AverageNs"])/1e3))
PY
bash $R/tools/pmc_any.sh scanany "scan_hits|hits_planes" tools/bench_scan.py --reads $READS --read_len $LEN --k $K --radius $RAD --reps 2
