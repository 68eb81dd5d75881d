#!/bin/bash
# SQ counters of the HIP 3x3 convolution at two UNet shapes: one rocprofv3 --kernel-trace --pmc pass per counter set (never combined
# with trace domains). Run on the GPU box from the repo root: bash tools/pmc_conv.sh > gpurun_out/pmc_conv.txt
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcc$i -o k -- python $R/tools/conv_bench.py --no-lib --no-check --iters 3 --only "320x320@64,1280x1280@16" > /tmp/pmcc$i.log 2>&1; grep -q hip_us /tmp/pmcc$i.log || tail -3 /tmp/pmcc$i.log
  python $R/tools/rocpd_stats.py --pmc /tmp/pmcc$i/k_results.db 2>/dev/null | grep -i "conv3x3_nhwc"
done
