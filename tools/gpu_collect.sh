#!/bin/bash
# Run on the GPU box (via gpurun): everything profiles/rNN holds for one tree, into gpurun_out/set_<tag>/ (copy into profiles/rNN/<tag>_*).
# Usage: tools/gpu_collect.sh <tag>
set -u
TAG=${1:-a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/set_$TAG
mkdir -p $OUT
cd $REPO
Q="--no-cpu-baseline --no-extra-workloads"
python3 bench.py > $OUT/bench_B1024.json 2> $OUT/bench_B1024.err || exit 1
echo "default line done"
python3 bench.py --workload c3 $Q > $OUT/bench_c3_B1024.json 2>> $OUT/err.log || exit 1
python3 bench.py --workload c4 --batch 256 --steps 10 --warmup 3 $Q > $OUT/bench_c4shapes_B256.json 2>> $OUT/err.log || exit 1
python3 bench.py --workload c4 --batch 256 --steps 10 --warmup 3 --dtype fp8 $Q > $OUT/bench_c5shapes_fp8_B256.json 2>> $OUT/err.log || exit 1
python3 bench.py --dtype fp8 $Q > $OUT/bench_fp8_B1024.json 2>> $OUT/err.log || exit 1
echo "bench lines done"
PMGT_BENCH_BACKEND=gloo PMGT_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 4 --steps 3 --warmup 1 --batch 256 $Q --no-end-to-end --no-batch-sweep > $OUT/bench_4rank_gloo_one_gpu_rehearsal.json 2>> $OUT/err.log || exit 1
echo "rehearsal done"
bash tools/gpu_profile.sh $TAG > $OUT/profile.log 2>&1 || exit 1
cp gpurun_out/prof_$TAG/kernel_stats.txt $OUT/kernel_stats.txt
cp gpurun_out/prof_$TAG/kt_bench.json $OUT/bench_under_rocprof.json
for f in gpurun_out/prof_$TAG/pmc_*.txt; do cp $f $OUT/$(basename $f | sed 's/pmc_fetch.txt/pmc_fetch_size.txt/; s/pmc_write.txt/pmc_write_size.txt/; s/pmc_fetch_6steps.txt/pmc_fetch_size_6steps.txt/; s/pmc_write_6steps.txt/pmc_write_size_6steps.txt/'); done
echo "c2 profile done"
BENCH_EXTRA="--workload c4 --batch 256" bash tools/gpu_profile.sh ${TAG}c4 > $OUT/profile_c4.log 2>&1 || exit 1
cp gpurun_out/prof_${TAG}c4/kernel_stats.txt $OUT/kernel_stats_c4shapes_B256.txt
for c in fetch write SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA; do cp gpurun_out/prof_${TAG}c4/pmc_$c.txt $OUT/c4shapes_pmc_$c.txt; done
echo "c4 profile done"
cd /tmp && export TMPDIR=/tmp
for B in 32 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$B -- python3 $REPO/bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-phase-profile --no-end-to-end --no-batch-sweep --no-extra-workloads > /dev/null 2>> $OUT/err.log
  python3 $REPO/tools/summarize_rocprof.py stats $OUT/kt$B > $OUT/kernel_stats_B$B.txt 2>&1
  rm -rf $OUT/kt$B
done
echo "all done"
