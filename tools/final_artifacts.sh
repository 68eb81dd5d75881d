set -x
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/xattn_fwd_hbm_traffic.json
timeout 300 python tools/pmc_traffic_kernel.py --imgs 64 > gpurun_out/pmc_traffic_kernel.txt 2>&1
timeout 300 python tools/pmc_traffic_kernel.py --imgs 64 --dtype bf16 >> gpurun_out/pmc_traffic_kernel.txt 2>&1
timeout 300 python tools/pmc_traffic_kernel.py --imgs 8 >> gpurun_out/pmc_traffic_kernel.txt 2>&1
cp gpurun_out/xattn_fwd_hbm_traffic.json profiles/xattn_fwd_hbm_traffic.json
timeout 900 bash tools/profile_bench.sh r05 > gpurun_out/profile_bench.log 2>&1
cp gpurun_out/prof_r05_kernel_stats.csv profiles/r05_bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT
timeout 1500 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log
tail -c 600 gpurun_out/bench_default.json
timeout 300 python -m pytest tests/test_kernel_gpu.py -q -m gpu -k "toolchain or optimistic or extreme" > gpurun_out/final_tests.txt 2>&1
tail -3 gpurun_out/final_tests.txt
