set -x
timeout 600 python -m pytest tests/test_fused_gpu.py -x -q -k "geglu or ff_" 2>&1 | tail -5
AB_ROUNDS=2 timeout 1200 python tools/lib_ab.py tools/proj_bench.py --imgs 32 --only ff1_fused --rounds 3 -- \
  base=src:sta_ffgemm.hip=build/ab/sta_ffgemm_base.hip \
  off= \
  mid=-DFF1_MID_BARRIER=1 \
  "odd=-DFF1_GROUP_B(wv)=((wv)&1)" \
  "oddmid=-DFF1_GROUP_B(wv)=((wv)&1),-DFF1_MID_BARRIER=1" \
  "none=-DFF1_GROUP_B(wv)=false" 2>&1 | grep -o '^\[[a-z]* r[0-9]\]\|"us": {[^}]*}' | paste - - 
