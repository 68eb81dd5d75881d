"""BASELINE configs[2] side leg of bench.py at a chosen prompt count / recomputation policy (tools only).
usage: wopt_probe.py <images> <policy> [find 0|1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from sta.pipeline import use_shipped_miopen_db
images, policy = int(sys.argv[1]), sys.argv[2]
find = len(sys.argv) > 3 and sys.argv[3] == "1"
use_shipped_miopen_db(0)
torch.backends.cudnn.benchmark = find
t0 = time.time()
r = bench.side_run(torch.device("cuda", 0), "fp16", 3, images, 1, 1, 512, 50, 2, checkpoint=policy, find=find)
r["wall_s"] = time.time() - t0
r["peak_gib"] = torch.cuda.max_memory_allocated() / 2 ** 30
print(json.dumps(r))
