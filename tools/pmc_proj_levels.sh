#!/bin/bash
# SQ counters and HBM-side traffic of the two projection-fused launches of a UNet call as bench.py issues them at 64 images:
# level 0 (head-pair kernel, query-fragment in, out-fragment out: tools/proj_bench.py --only pairqo) and level 1 (locals-from-L2 kernel,
# query-fragment in: --N 1024 --C 640 --only pairq). One rocprofv3 --kernel-trace --pmc pass per counter set (never combined with other
# trace domains). GPU box, from the repo root:   bash tools/pmc_proj_levels.sh > gpurun_out/pmc_proj_levels.txt
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
run() {   # tag, proj_bench arguments
  tag=$1; shift
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf /tmp/pmcl_$tag$i
    rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcl_$tag$i -o k -- python $R/tools/proj_bench.py --imgs 64 --iters 10 --rounds 1 "$@" > /tmp/pmcl_$tag$i.log 2>&1
    python $R/tools/rocpd_stats.py --pmc /tmp/pmcl_$tag$i/k_results.db 2>/dev/null | grep -i "xattn_fwd_proj" | sed "s/^/$tag /"
  done
}
run level0_fp16 --only pairqo
run level1_fp16 --N 1024 --C 640 --only pairq
run level0_bf16 --only pairqo --dtype bf16
