#!/bin/bash
# SQ counters of the self-attention forward kernels (scaled and log2-domain paths) at the bench shapes: one rocprofv3
# --kernel-trace --pmc pass per counter set (never combined with trace domains). Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcs$i -o k -- python $R/tools/selfattn_fwd_bench.py > /tmp/pmcs$i.log 2>&1; grep -q log2_us /tmp/pmcs$i.log || tail -3 /tmp/pmcs$i.log
  python $R/tools/rocpd_stats.py --pmc /tmp/pmcs$i/k_results.db 2>/dev/null | grep -i "selfattn_fwd_kernelIDF16_Li2ELi3\|selfattn_fwd32_kernelIDF16_"
done
