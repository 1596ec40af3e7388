#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + HBM-traffic counters of the default bench command.
# Usage: tools/gpu_profile.sh <tag>   -> gpurun_out/prof_<tag>/{kernel_stats.txt, pmc_*.txt}
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-phase-profile --no-end-to-end --no-batch-sweep --no-extra-workloads ${BENCH_EXTRA:-}"
PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-phase-profile --no-end-to-end --no-batch-sweep --no-extra-workloads ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py $BENCH_ARGS > $OUT/kt_bench.json 2> $OUT/kt.err
python3 $REPO/tools/summarize_rocprof.py stats $OUT/kt > $OUT/kernel_stats.txt 2>&1
# counters in their own passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/pmc_write.err
python3 $REPO/tools/summarize_rocprof.py pmc $OUT/pmc_fetch FETCH_SIZE > $OUT/pmc_fetch.txt 2>&1
python3 $REPO/tools/summarize_rocprof.py pmc $OUT/pmc_write WRITE_SIZE > $OUT/pmc_write.txt 2>&1
# the same two runs dispatch by dispatch (order = launch order), and the phase order of one step of the same configuration
python3 $REPO/tools/summarize_rocprof.py pmcseq $OUT/pmc_fetch FETCH_SIZE > $OUT/pmc_fetch_seq.txt 2>&1
python3 $REPO/tools/summarize_rocprof.py pmcseq $OUT/pmc_write WRITE_SIZE > $OUT/pmc_write_seq.txt 2>&1
python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-batch-sweep --no-extra-workloads ${BENCH_EXTRA:-} --detail-out $OUT/detail.json > /dev/null 2> $OUT/detail.err
# the same two counters over SIX steps: the whole-step traffic is the DIFFERENCE of the two runs divided by the three extra steps
# (tools/make_traffic.py), so that set-up kernels -- launched once, or a multiple of the step count -- never count as step traffic
PMC_ARGS_B="--steps 5 --warmup 1 --no-cpu-baseline --no-phase-profile --no-end-to-end --no-batch-sweep --no-extra-workloads ${BENCH_EXTRA:-}"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_b -- python3 $REPO/bench.py $PMC_ARGS_B > /dev/null 2> $OUT/pmc_fetch_b.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_b -- python3 $REPO/bench.py $PMC_ARGS_B > /dev/null 2> $OUT/pmc_write_b.err
python3 $REPO/tools/summarize_rocprof.py pmc $OUT/pmc_fetch_b FETCH_SIZE > $OUT/pmc_fetch_6steps.txt 2>&1
python3 $REPO/tools/summarize_rocprof.py pmc $OUT/pmc_write_b WRITE_SIZE > $OUT/pmc_write_6steps.txt 2>&1
rm -rf $OUT/pmc_fetch_b $OUT/pmc_write_b
# matrix-pipe and vector-ALU occupancy per kernel over the same step (SQ counters, one group per pass):
#   matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES); VALU / MFMA instruction counts
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/pmc_sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/pmc_sq2.err
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES; do python3 $REPO/tools/summarize_rocprof.py pmc $OUT/pmc_sq1 $c > $OUT/pmc_$c.txt 2>&1; done
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU; do python3 $REPO/tools/summarize_rocprof.py pmc $OUT/pmc_sq2 $c > $OUT/pmc_$c.txt 2>&1; done
# LDS array cycles and the extra cycles bank conflicts add (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = share of LDS cycles lost to conflicts)
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq3 -- python3 $REPO/bench.py $PMC_ARGS > /dev/null 2> $OUT/pmc_sq3.err
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python3 $REPO/tools/summarize_rocprof.py pmc $OUT/pmc_sq3 $c > $OUT/pmc_$c.txt 2>&1; done
rm -rf $OUT/pmc_sq3
# keep only the summaries (raw traces are large)
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2
tail -n 45 $OUT/kernel_stats.txt
