#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun; delete the LOCAL gpurun_out/round_prof and
# gpurun_out/pmc_round first - gpurun merges into them and stale per-process CSVs would be averaged in); summaries land in gpurun_out/ and are
# copied into profiles/ by tools/collect_profiles.py.  Kernel trace and PMC passes are separate runs.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round_prof
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py --no-cpu-baseline --no-c4 --no-stages --no-embed-dist --shard-proxy 0 --e2e none --steps 20 --warmup 5 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bench_pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-c4 --no-c5 --no-stages --no-embed-dist --shard-proxy 0 --e2e none --steps 5 --warmup 1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bench_pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-c4 --no-c5 --no-stages --no-embed-dist --shard-proxy 0 --e2e none --steps 5 --warmup 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e_trace -- python3 $R/tools/e2e.py --config C3 --mode fast --iters 100 > $OUT/e2e_trace.json 2> $OUT/e2e_trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/embed_trace -- python3 $R/tools/bench_embed.py --n 50000 --iters 20 > $OUT/embed_trace.txt 2> $OUT/embed_trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/seqshard_trace -- python3 $R/tools/probes/seq_shard_only.py adder > $OUT/seqshard_trace.txt 2> $OUT/seqshard_trace.err
# (the full-size C5 scan is part of bench_trace: bench.py's c5 leg generates its 50 M x 300 bp reads in HBM)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scan_trace -- python3 $R/tools/bench_scan.py --reads 10000000 --read_len 150 --k 8 --radius 2 --reps 5 > $OUT/scan_trace.json 2> $OUT/scan_trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/count_trace -- python3 $R/tools/probes/count_only.py 14 5 0 > $OUT/count_trace.txt 2> $OUT/count_trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/keyspace_trace -- python3 $R/tools/probes/keyspace_one.py 14 8 3 > $OUT/keyspace_trace.txt 2> $OUT/keyspace_trace.err
# counters for every kernel DESIGN.md calls "bound by ..." (C3 pipeline + SEQ micro-benchmark, four PMC passes)
bash $R/tools/pmc_round.sh round > $OUT/pmc_round.log 2>&1 || true
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_round $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1 || true
ls $OUT
