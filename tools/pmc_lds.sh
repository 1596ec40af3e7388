#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-phase-profile --no-end-to-end --no-batch-sweep"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/p1.err
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES; do python3 $REPO/tools/summarize_rocprof.py pmc $OUT/p1 $c > $OUT/pmc_$c.txt 2>&1; done
rm -rf $OUT/p1
