cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcs$i -o k -- python $R/tools/selfattn_bench.py 16 5 > /tmp/pmcs$i.log 2>&1 || tail -3 /tmp/pmcs$i.log
  python $R/tools/rocpd_stats.py --pmc /tmp/pmcs$i/k_results.db 2>/dev/null | grep -i "selfattn" | grep ",65536," | awk -F, "{print \$(NF-2), \$(NF-1), \$NF}"
done
