#!/bin/bash
# What bounds spconv_wgrad_k?  PMC passes over tools/prof_wgrad.py (one level, one batch): SQ issue / wait and instruction counts,
# texture-addresser and vector-L1 stalls, L2 hit rate.  Separate passes (kernel-trace only), per-kernel means printed.
# usage (GPU box): bash tools/pmc_wgrad.sh [level=1] [batch=8]      (~15 s per pass)
LV=${1:-1}; B=${2:-8}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_wgrad; mkdir -p $OUT
CMD="python $R/tools/prof_wgrad.py $B $LV fp32"
export FILTER="wgrad_k|spconv_gmm"
{
echo "== level $LV batch $B"
bash $R/tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD" $CMD
bash $R/tools/pmc_pass.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" $CMD
bash $R/tools/pmc_pass.sh "TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" $CMD
bash $R/tools/pmc_pass.sh "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" $CMD
bash $R/tools/pmc_pass.sh "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" $CMD
bash $R/tools/pmc_pass.sh "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" $CMD
} 2>&1 | tee $OUT/level${LV}_b${B}.txt
