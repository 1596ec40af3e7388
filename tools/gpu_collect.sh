#!/bin/bash
# Run on the GPU box (via gpurun): everything profiles/rNN holds for one tree, into gpurun_out/set_<tag>/ (copy into profiles/rNN/<tag>_*).
# Usage: tools/gpu_collect.sh <tag> [lines|c2|c4|c4i|i4d|small]   (one part per gpurun call: the whole set is longer than one call's limit)
set -u
TAG=${1:-a}
PART=${2:-lines}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/set_$TAG
mkdir -p $OUT
cd $REPO
Q="--no-cpu-baseline --no-extra-workloads"
QQ="$Q --no-end-to-end --no-batch-sweep"
if [ "$PART" = "lines" ]; then
  # stdout = the ONE compact line the driver records (<name>.line.json), --detail-out = the full record (<name>.json)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $OUT/bench_B1024.json > $OUT/bench_B1024.line.json 2> /dev/null || exit 1
  echo "default line done: $(wc -c < $OUT/bench_B1024.line.json) bytes"
  python3 bench.py --workload c3 $Q --detail-out $OUT/bench_c3_B1024.json > $OUT/bench_c3_B1024.line.json 2>> $OUT/err.log || exit 1
  python3 bench.py --dtype fp8 $QQ --detail-out $OUT/bench_fp8_B1024.json > /dev/null 2>> $OUT/err.log || exit 1
  python3 bench.py --force-exchange --buckets two $QQ --detail-out $OUT/bench_rccl_one_rank_two_buckets.json > /dev/null 2>> $OUT/err.log || exit 1
  python3 bench.py --force-exchange --buckets layer $QQ --detail-out $OUT/bench_rccl_one_rank_layer_buckets.json > /dev/null 2>> $OUT/err.log || exit 1
  python3 bench.py --force-exchange --buckets one $QQ --detail-out $OUT/bench_rccl_one_rank_one_bucket.json > /dev/null 2>> $OUT/err.log || exit 1
  python3 bench.py $QQ --detail-out $OUT/bench_plain_after_rccl.json > /dev/null 2>> $OUT/err.log || exit 1
  PMGT_BENCH_BACKEND=gloo PMGT_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 4 --steps 3 --warmup 1 --batch 256 $QQ --detail-out $OUT/bench_4rank_gloo_one_gpu_rehearsal.json > $OUT/bench_4rank_gloo_one_gpu_rehearsal.line.json 2>> $OUT/err.log || exit 1
  echo "lines done"
fi
copy_set() {      # copy_set <prof tag> <prefix>
  cp gpurun_out/prof_$1/kernel_stats.txt $OUT/$2kernel_stats.txt
  cp gpurun_out/prof_$1/kt_bench.json $OUT/$2bench_under_rocprof.json
  cp gpurun_out/prof_$1/detail.json $OUT/$2detail.json
  for f in gpurun_out/prof_$1/pmc_*.txt; do cp $f $OUT/$2$(basename $f | sed 's/pmc_fetch.txt/pmc_fetch_size.txt/; s/pmc_write.txt/pmc_write_size.txt/; s/pmc_fetch_6steps.txt/pmc_fetch_size_6steps.txt/; s/pmc_write_6steps.txt/pmc_write_size_6steps.txt/'); done
}
if [ "$PART" = "c2" ]; then
  bash tools/gpu_profile.sh $TAG > $OUT/profile.log 2>&1 || exit 1
  copy_set $TAG ""
  echo "c2 profile done"
fi
if [ "$PART" = "i4d" ]; then
  BENCH_EXTRA="--intermediate 1024" bash tools/gpu_profile.sh ${TAG}i1024 > $OUT/profile_i1024.log 2>&1 || exit 1
  copy_set ${TAG}i1024 "c2_i1024_"
  BENCH_EXTRA="--beta 1.0" bash tools/gpu_profile.sh ${TAG}beta1 > $OUT/profile_beta1.log 2>&1 || exit 1
  copy_set ${TAG}beta1 "c2_beta1_"
  echo "i = 4d / beta = 1 profiles done"
fi
if [ "$PART" = "c4" ]; then
  BENCH_EXTRA="--workload c4 --batch 256" bash tools/gpu_profile.sh ${TAG}c4 > $OUT/profile_c4.log 2>&1 || exit 1
  copy_set ${TAG}c4 "c4_"
  echo "c4 profile done"
fi
if [ "$PART" = "c4b" ]; then
  BENCH_EXTRA="--workload c4 --batch 1024" bash tools/gpu_profile.sh ${TAG}c4b1024 > $OUT/profile_c4_b1024.log 2>&1 || exit 1
  copy_set ${TAG}c4b1024 "c4_b1024_"
  echo "c4 B = 1024 profile done"
fi
if [ "$PART" = "c4i" ]; then
  BENCH_EXTRA="--workload c4 --batch 256 --intermediate 2048" bash tools/gpu_profile.sh ${TAG}c4i2048 > $OUT/profile_c4_i2048.log 2>&1 || exit 1
  copy_set ${TAG}c4i2048 "c4_i2048_"
  echo "c4 i = 4d profile done"
fi
if [ "$PART" = "small" ]; then
  cd /tmp && export TMPDIR=/tmp
  for B in 32 256; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$B -- python3 $REPO/bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-phase-profile --no-end-to-end --no-batch-sweep --no-extra-workloads > /dev/null 2>> $OUT/err.log
    python3 $REPO/tools/summarize_rocprof.py stats $OUT/kt$B > $OUT/kernel_stats_B$B.txt 2>&1
    rm -rf $OUT/kt$B
  done
  echo "small batches done"
fi
echo "part $PART done"
