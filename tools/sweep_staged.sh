for lvl in 1 2 3; do
  for w in 0 4 8; do
    for t in 0 1 2 4 6 8 12; do
      echo -n "L$lvl waves=$w tiles=$t  "
      python tools/kernel_bench.py --imgs 32 --iters 60 --level $lvl --dtype fp16 --kernel 1 --opt 2=$w --opt 1=$t 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['us'], d['GBps'])"
    done
  done
  echo -n "L$lvl split  "; python tools/kernel_bench.py --imgs 32 --iters 60 --level $lvl --dtype fp16 --kernel 2 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['us'], d['GBps'])"
done
