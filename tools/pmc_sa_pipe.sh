#!/bin/bash
# SQ counters of the level-0 self-attention forward launch, software-pipelined loop (default) and plain loop (STA_SA_MODE=2): one
# rocprofv3 --kernel-trace --pmc pass per counter set (never combined with other trace domains). GPU box, from the repo root.
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in 0 2; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    STA_SA_MODE=$mode rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcp$mode$i -o k -- python $R/tools/selfattn_l0_time.py > /tmp/pmcp$mode$i.log 2>&1
    python $R/tools/rocpd_stats.py --pmc /tmp/pmcp$mode$i/k_results.db 2>/dev/null | grep -i "selfattn_fwd" | sed "s/^/mode$mode /"
  done
done
