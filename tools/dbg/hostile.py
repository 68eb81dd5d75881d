import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
import bench
for dt in ("fp16", "bf16"):
    print(json.dumps(bench.hostile_logits_leg(torch.device("cuda", 0), dt, 64, 2, 64), indent=1))
