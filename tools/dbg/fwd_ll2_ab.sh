#!/bin/bash
# forward cross-attention at levels 1 (K = 4), 2, mid: locals from L2 (default) vs grouped staging (option 10 = 2), same process order alternated
for imgs in 64 16; do for lv in 2 3; do for o in 0 2 0 2; do
  echo -n "imgs=$imgs level=$lv opt10=$o: "; python tools/kernel_bench.py --dtype fp16 --imgs $imgs --level $lv --opt 10=$o 2>&1 | grep -o '"us": [0-9.]*'
done; done; done
for o in 0 2 0 2; do echo -n "imgs=16 level=1 K=4 opt10=$o: "; python tools/kernel_bench.py --dtype fp16 --imgs 16 --level 1 --K 4 --opt 10=$o 2>&1 | grep -o '"us": [0-9.]*'; done
