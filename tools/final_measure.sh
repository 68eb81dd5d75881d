#!/bin/bash
# Round-end measurements on the final tree (GPU box, repo root; everything lands in gpurun_out/, copy what is quoted to profiles/):
#   1. PMC traffic keyed to the kernel sources — the level-0 forward (both 16-bit types at the bench's 64 images per launch, and the 8-image
#      launch of configs[3]'s per-GPU shard) and the backward at the four levels (bench.py refuses a traffic figure of other sources)
#   2. rocprofv3 kernel traces of the default bench command and of BASELINE configs[2] at 16 prompts per step
#   3. the bench line itself, un-profiled, with every side leg
# usage: bash tools/final_measure.sh r06
tag=${1:-r06}
for args in "--imgs 64 --dtype fp16" "--imgs 64 --dtype bf16" "--imgs 8 --dtype fp16"; do
  timeout 300 python tools/pmc_traffic_kernel.py $args 2>&1 | tail -2
done
timeout 500 python tools/pmc_traffic_bwd.py 2>&1 | tail -6
cp gpurun_out/xattn_fwd_hbm_traffic.json gpurun_out/xattn_bwd_hbm_traffic.json profiles/      # on the box: the bench below reads them
PROF_TIMEOUT=900 bash tools/profile_bench.sh ${tag}_bench > gpurun_out/pb1.log 2>&1
PROF_TIMEOUT=1100 bash tools/profile_bench.sh ${tag}_wopt16 --opt-epochs 3 --images-per-step 16 --steps 1 --warmup 1 > gpurun_out/pb2.log 2>&1
timeout 900 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.log
tail -c 400 gpurun_out/${tag}_bench_default.json
