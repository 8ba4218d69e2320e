cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_scan -- python3 $R/tools/bench_scan.py --reps 5 > $R/gpurun_out/prof_scan.json 2> $R/gpurun_out/prof_scan.err
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_scan/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), ("%.3f ms" % (float(r["AverageNs"]) / 1e6)).rjust(12))
PY
