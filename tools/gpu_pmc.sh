#!/bin/bash
# Run on the GPU box (via gpurun): one rocprofv3 counter pass over a short bench run.
# Usage: BENCH_EXTRA="..." tools/gpu_pmc.sh <tag> COUNTER [COUNTER ...]   -> gpurun_out/pmc_<tag>/pmc_<COUNTER>.txt
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-phase-profile --no-end-to-end --no-batch-sweep ${BENCH_EXTRA:-}"
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/raw -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/pmc.err
for c in "$@"; do python3 $REPO/tools/summarize_rocprof.py pmc $OUT/raw $c > $OUT/pmc_$c.txt 2>&1; done
rm -rf $OUT/raw
