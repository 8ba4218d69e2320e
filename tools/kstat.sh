#!/bin/bash
# kernel-time summary (rocprofv3 --kernel-trace --stats) of a python command; prints kernels matching a pattern
#   usage: bash tools/kstat.sh <pattern> <script.py> [args...]      (env assignments are inherited)
cd /tmp && export TMPDIR=/tmp
PAT=$1; shift
OUT=$(mktemp -d /tmp/kstat.XXXXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/"$@" > $OUT/out.txt 2> $OUT/err.txt || tail -5 $OUT/err.txt
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Name"]:
            print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"])/1e6:8.3f} ms')
PY
rm -rf $OUT
