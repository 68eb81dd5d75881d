#!/bin/bash
# SQ / TA / TCC counters of the projection-fused forward kernel at the bench launch (level 0, 16 images): one
# rocprofv3 --kernel-trace --pmc pass per counter set (never combined with trace domains). Run on the GPU box from
# the repo root; prints one line per set.  Usage: tools/pmc_proj.sh [extra proj_bench.py args]
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcp$i -o k -- python $R/tools/proj_bench.py --only ${ARM:-pair} --iters 10 --rounds 1 "$@" > /tmp/pmcp$i.log 2>&1 || tail -3 /tmp/pmcp$i.log
  python $R/tools/rocpd_stats.py --pmc /tmp/pmcp$i/k_results.db 2>/dev/null | grep -i "proj_pair_kernel\|proj_kernel\|proj_p3"
done
