#!/bin/bash
# HBM-side traffic of the fused forward kernel per UNet level: separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE)
# over tools/kernel_bench.py, one level and one images-per-launch value at a time; writes
# gpurun_out/xattn_fwd_hbm_traffic.json (copy to profiles/). Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmct && mkdir -p /tmp/pmct
for I in 1 8 16; do for L in 0 1 2 3; do for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $CNT -d /tmp/pmct/${I}_${L}_${CNT} -o k -- python $R/tools/kernel_bench.py --iters 20 --imgs $I --level $L > /tmp/pmct/log_${I}_${L}_${CNT}.txt 2>&1
done; done; done
python $R/tools/pmc_traffic_json.py /tmp/pmct > $R/gpurun_out/xattn_fwd_hbm_traffic.json
