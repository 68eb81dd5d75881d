# headline workload with the optimistic self-attention on / off (STA_SELFATTN_OPTIMISTIC=0), same box, fp16
cd $GRAFT_REPO_ROOT
for o in 1 0 1 0; do
STA_SELFATTN_OPTIMISTIC=$o timeout 600 python bench.py --steps 2 --warmup 1 --no-side-runs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('optimistic=$o', round(d['value'],3), 'images/s', d['config'].get('selfattn_optimistic'))"
done
