#!/bin/bash
# HBM-side traffic of the HIP 3x3 convolution (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes, kernel trace only) at two
# UNet shapes; units per MI355X_MICROARCH.md: FETCH_SIZE in 64-byte... reported raw here, converted in profiles/r04_conv3x3.md.
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmct_$c -o k -- python $R/tools/conv_bench.py --no-lib --no-check --iters 3 --only "320x320@64,1280x1280@16" > /tmp/pmct_$c.log 2>&1
  python $R/tools/rocpd_stats.py --pmc /tmp/pmct_$c/k_results.db 2>/dev/null | grep -i "conv3x3_nhwc"
done
