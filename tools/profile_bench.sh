#!/bin/bash
# Round-N profile of the default bench under rocprofv3 (kernel trace only; no PMC here): writes
#   gpurun_out/prof_$1_kernel_stats.csv, prof_$1_breakdown.txt, prof_$1_bench.json   (copy to profiles/ as $1_*)
# usage (GPU box, repo root): bash tools/profile_bench.sh r02 [extra bench.py args]
tag=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout ${PROF_TIMEOUT:-1500} rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o k -- python $R/bench.py --no-cpu-baseline --no-side-runs --no-hostile --steps 2 --no-graph "$@" \
  > $R/gpurun_out/prof_${tag}_bench.json 2> $R/gpurun_out/prof_${tag}_run.log
db=$(ls /tmp/prof_$tag/*results.db /tmp/prof_$tag/*/*results.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $db > $R/gpurun_out/prof_${tag}_kernel_stats.csv
python $R/tools/rocpd_stats.py --breakdown $db 16 > $R/gpurun_out/prof_${tag}_breakdown.txt
grep -i "xattn\|selfattn" $R/gpurun_out/prof_${tag}_kernel_stats.csv | head -12
cat $R/gpurun_out/prof_${tag}_breakdown.txt | head -14
