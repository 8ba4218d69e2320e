#!/bin/bash
# PMC passes over the embedding micro-benchmark only (forces kernels): bash tools/pmc_embed.sh <tag> [bench_embed args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
OUT=$R/gpurun_out/pmc_embed_$TAG
rm -rf $OUT && mkdir -p $OUT
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
rocprofv3 --pmc $A --output-format csv -d $OUT/e2e_sq_a -- python3 $R/tools/bench_embed.py "$@" > $OUT/a.txt 2> $OUT/a.err
rocprofv3 --pmc $B --output-format csv -d $OUT/e2e_sq_b -- python3 $R/tools/bench_embed.py "$@" > $OUT/b.txt 2> $OUT/b.err
find $OUT -type f ! -name '*counter_collection.csv' ! -name '*.txt' -delete
python3 $R/tools/pmc_summary.py $OUT $OUT/summary.json | grep -E "kernel|forces|sym_reduce|apply|tail"
