# level 0 in fp16: the optimistic softmax (product build) against -DSTA_P3_OPTIMISTIC=0, the headline workload + its in-situ roofline leg, same box
# (build/ab/libsta_{opt,std}.so are built beforehand, GPU-less: AB_ROUNDS=0 python tools/lib_ab.py bench.py --steps 1 -- opt= std=-DSTA_P3_OPTIMISTIC=0)
cd $GRAFT_REPO_ROOT
AB_ROUNDS=2 python tools/lib_ab.py bench.py --steps 1 --warmup 1 --no-side-runs --no-cpu-baseline -- opt= std=-DSTA_P3_OPTIMISTIC=0 2>/dev/null | python -c "
import json, sys
for line in sys.stdin:
    tag, _, rest = line.partition('] ')
    try: d = json.loads(rest)
    except Exception: continue
    r = d['roofline']
    print(tag + ']', 'images/s %.3f' % d['value'], 'level-0 launch in situ %.1f us' % r['avg_launch_us'], 'frac %.4f' % r['frac'], 'mfma %.4f' % r['mfma_frac'], 'all 16 launches:', {k: round(v, 4) for k, v in r['all_launches'].items() if isinstance(v, float)})
"
