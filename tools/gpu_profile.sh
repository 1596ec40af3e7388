#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + HBM-traffic counters of the default bench command.
# Usage: tools/gpu_profile.sh <tag>   -> gpurun_out/prof_<tag>/{kernel_stats.txt, pmc_*.txt}
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-phase-profile ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py $BENCH_ARGS > $OUT/kt_bench.json 2> $OUT/kt.err
python3 $REPO/tools/summarize_rocprof.py stats $OUT/kt > $OUT/kernel_stats.txt 2>&1
# counters in their own passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-phase-profile ${BENCH_EXTRA:-} > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-phase-profile ${BENCH_EXTRA:-} > /dev/null 2> $OUT/pmc_write.err
python3 $REPO/tools/summarize_rocprof.py pmc $OUT/pmc_fetch FETCH_SIZE > $OUT/pmc_fetch.txt 2>&1
python3 $REPO/tools/summarize_rocprof.py pmc $OUT/pmc_write WRITE_SIZE > $OUT/pmc_write.txt 2>&1
# keep only the summaries (raw traces are large)
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_write
tail -n 45 $OUT/kernel_stats.txt
