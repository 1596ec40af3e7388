#!/bin/bash
# HBM traffic counters for the op-level microbenchmarks (one kernel family per run)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/pmc_micro; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $REPO/tools/microbench.py 1024 > /dev/null 2> $OUT/$c.err
  python3 $REPO/tools/summarize_rocprof.py pmc $OUT/$c $c > $OUT/$c.txt 2>&1
  rm -rf $OUT/$c
done
head -8 $OUT/FETCH_SIZE.txt | cut -c1-70,112-150; head -8 $OUT/WRITE_SIZE.txt | cut -c1-70,112-150
