set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round_prof
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py --no-cpu-baseline --no-c4 --no-stages --no-embed-dist --shard-proxy 0 --e2e none --steps 20 --warmup 5 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bench_pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-c4 --no-c5 --no-stages --no-embed-dist --shard-proxy 0 --e2e none --steps 5 --warmup 1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bench_pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-c4 --no-c5 --no-stages --no-embed-dist --shard-proxy 0 --e2e none --steps 5 --warmup 1 > /dev/null 2> $OUT/pmc_fetch.err
ls $OUT
