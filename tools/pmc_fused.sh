#!/bin/bash
# SQ / LDS counters of the fused projection + attention forward (microbench), one counter group per pass
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/pmc_fused; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
run() {  # name, counters
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $REPO/tools/microbench.py 1024 ${MODE:-fused} > /dev/null 2> $OUT/$1.err
  for c in $2; do python3 $REPO/tools/summarize_rocprof.py pmc $OUT/$1 $c 2>/dev/null | grep -E "${KPAT:-qkvc_attn_fwd}|kernel " | head -3; done
  rm -rf $OUT/$1
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
run d "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"
run e "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM"
