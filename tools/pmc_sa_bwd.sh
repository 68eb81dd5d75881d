#!/bin/bash
# SQ counters of the level-0 self-attention backward kernels (B = 32, N = 4096, d = 40, eight-wave workgroups, fp16 + bf16 in one process).
# One rocprofv3 --kernel-trace --pmc pass per counter set (never combined with other trace domains). GPU box, from the repo root.
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmcsb$i
  SA_PRE=1 timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcsb$i -o k -- python $R/tools/dbg/sa_bwd_waves_ab.py > /tmp/pmcsb$i.log 2>&1
  db=$(ls /tmp/pmcsb$i/*results.db /tmp/pmcsb$i/*/*results.db 2>/dev/null | head -1)
  python $R/tools/rocpd_stats.py --pmc $db 2>/dev/null | grep "selfattn_bwd_d.*IDF16_Li2ELi3ELi2ELi8" | cut -c14-200
done
