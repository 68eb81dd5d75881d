set -x
AB_ROUNDS=2 timeout 1500 python tools/lib_ab.py tools/conv_bench.py --no-lib --no-check --only "320x320@64,960x320@64,640x640@32,1280x1280@16,1280x1280@8" -- "$@" 2>&1 | grep -v "per_unet_call\|amdgpu.ids"
