#!/bin/bash
# Round-end measurements that are keyed to the kernel sources (GPU box, repo root): PMC traffic of the level-0 forward (both 16-bit types,
# the bench's 64 images per launch and the 8-image launch of the tracked epochs' fixed-weight trajectory) and of the backward at the four
# levels. Results land in gpurun_out/*.json: copy to profiles/.
for args in "--imgs 64 --dtype fp16" "--imgs 64 --dtype bf16" "--imgs 8 --dtype fp16"; do
  timeout 300 python tools/pmc_traffic_kernel.py $args 2>&1 | tail -2
done
timeout 500 python tools/pmc_traffic_bwd.py 2>&1 | tail -6
