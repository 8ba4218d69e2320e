cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/tools/e2e.py --config C3 --mode fast --iters 50 > $R/gpurun_out/prof_c3.json 2> $R/gpurun_out/prof_c3.err
tail -1 $R/gpurun_out/prof_c3.json | cut -c1-1500
cut -c1-110,200-330 $R/gpurun_out/prof_c3/*/*kernel_stats.csv | head -24
