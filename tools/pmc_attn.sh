#!/bin/bash
# SQ wait/issue counters of the attention kernels (microbench), one counter group per pass
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/pmc_attn; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
run() {  # name, counters, mode
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $REPO/tools/microbench.py 1024 $3 > /dev/null 2> $OUT/$1.err
  for c in $2; do python3 $REPO/tools/summarize_rocprof.py pmc $OUT/$1 $c 2>/dev/null | grep -E "attn_bwd_mfma|qkvc_attn_fwd|attn_fwd_mfma|kernel " | head -4; done
  rm -rf $OUT/$1
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" attn
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" attn
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" attn
run d "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" fused
