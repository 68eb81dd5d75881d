#!/bin/bash
# per-kernel durations of the self-attention backward at level 0 (B = 32): log2-domain q (PRE kernels) against scale = d^-1/2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for pre in 1 0; do
  rm -rf /tmp/sab_$pre
  SA_PRE=$pre timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sab_$pre -o k -- python $R/tools/dbg/sa_bwd_waves_ab.py > /dev/null 2>&1
  db=$(ls /tmp/sab_$pre/*results.db /tmp/sab_$pre/*/*results.db 2>/dev/null | head -1)
  echo "== SA_PRE=$pre"; python $R/tools/rocpd_stats.py $db | head -12 | cut -c1-150
done
