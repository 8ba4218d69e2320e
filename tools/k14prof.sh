cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/k14prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/probes/count_only.py 14 5 0 > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/t/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        print("%-50s %4s %9.1f us" % (r["Name"].split("::")[-1][:50], r["Calls"], float(r["AverageNs"])/1e3))
PY
