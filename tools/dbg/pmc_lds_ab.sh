cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
for tag in swz noswz; do
  rm -rf /tmp/pl_$tag
  STA_LIB_OVERRIDE=$R/build/ab/libsta_$tag.so timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pl_$tag -o k -- python -c "import sys; sys.argv=['proj_bench.py','--only','pairqo','--imgs','64','--iters','10','--rounds','1']; sys.path.insert(0,'$R/diffusion-spacetime-attn_amd'); from sta import lib; lib.LIB_PATH='$R/build/ab/libsta_$tag.so'; __file__='$R/tools/proj_bench.py'; exec(open(__file__).read())" > /tmp/pl_$tag.log 2>&1
  python $R/tools/rocpd_stats.py --pmc /tmp/pl_$tag/k_results.db 2>/dev/null | grep -i "xattn_fwd_proj" | sed "s/^/$tag /"
done
