#!/bin/bash
# SQ counters of the level-0 self-attention forward launch in fp16: the optimistic loop (STA_SA_OPT=1: key loop from the own block, no running
# maximum) against the standard loop (STA_SA_OPT=0), logits that flag nothing (STA_SA_QSCALE=0.25). One rocprofv3 --kernel-trace --pmc pass per
# counter set (never combined with other trace domains). GPU box, from the repo root.
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
for opt in 1 0; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    STA_SA_QSCALE=0.25 STA_SA_OPT=$opt rocprofv3 --kernel-trace --pmc $set -d /tmp/pmco$opt$i -o k -- python $R/tools/selfattn_l0_time.py > /tmp/pmco$opt$i.log 2>&1
    python $R/tools/rocpd_stats.py --pmc /tmp/pmco$opt$i/k_results.db 2>/dev/null | grep -i "selfattn_fwd" | sed "s/^/opt$opt /"
  done
done
