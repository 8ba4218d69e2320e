#!/bin/bash
# PMC evidence for the kernels DESIGN.md calls "bound by ..." (VERDICT r01 item 3): separate rocprofv3 --pmc passes (never
# combined with a trace), the program directly after `--`.  Two workloads cover every kernel: the C3 pipeline (count, dedupe,
# mask, scan, knn select / sums, symmetric FAST forces) and the embedding micro-benchmark in SEQ mode.
#   usage (through gpurun):  bash tools/pmc_round.sh [tag]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_${1:-r03}
rm -rf $OUT && mkdir -p $OUT
declare -A SETS
SETS[sq_a]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SETS[sq_b]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
SETS[fetch]="FETCH_SIZE"
SETS[write]="WRITE_SIZE"
for s in sq_a sq_b fetch write; do
  echo "== pass $s: ${SETS[$s]}"
  rocprofv3 --pmc ${SETS[$s]} --output-format csv -d $OUT/e2e_$s -- python3 $R/tools/e2e.py --config C3 --mode fast --iters 12 > $OUT/e2e_$s.json 2> $OUT/e2e_$s.err || { tail -5 $OUT/e2e_$s.err; echo "pass $s (e2e) failed"; }
  rocprofv3 --pmc ${SETS[$s]} --output-format csv -d $OUT/count14_$s -- python3 $R/tools/probes/count_only.py 14 3 0 > $OUT/count14_$s.txt 2> $OUT/count14_$s.err || { tail -5 $OUT/count14_$s.err; echo "pass $s (count14) failed"; }
  rocprofv3 --pmc ${SETS[$s]} --output-format csv -d $OUT/keyspace14_$s -- python3 $R/tools/probes/keyspace_one.py 14 8 3 > $OUT/keyspace14_$s.txt 2> $OUT/keyspace14_$s.err || { tail -5 $OUT/keyspace14_$s.err; echo "pass $s (keyspace14) failed"; }
  rocprofv3 --pmc ${SETS[$s]} --output-format csv -d $OUT/seq_$s -- python3 $R/tools/bench_embed.py --n 50000 --iters 3 --modes seq > $OUT/seq_$s.txt 2> $OUT/seq_$s.err || { tail -5 $OUT/seq_$s.err; echo "pass $s (seq) failed"; }
  rocprofv3 --pmc ${SETS[$s]} --output-format csv -d $OUT/seqshard_$s -- python3 $R/tools/probes/seq_shard_only.py adder > $OUT/seqshard_$s.txt 2> $OUT/seqshard_$s.err || { tail -5 $OUT/seqshard_$s.err; echo "pass $s (seqshard) failed"; }
done
# keep only the counter CSVs (the merge-back is capped at 64 MiB)
find $OUT -type f ! -name '*counter_collection.csv' ! -name '*.json' ! -name '*.txt' ! -name '*.err' -delete
du -sh $OUT
