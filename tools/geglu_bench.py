"""sta_geglu at the FF shapes of the default bench (32 prompts: rows = 64 * N): time, HBM-side GB/s and max error vs fp64."""
import json
import sys
import os
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import fused  # noqa: E402

for R, D in ((64 * 4096, 1280), (64 * 1024, 2560), (64 * 256, 5120)):
    h = (torch.randn(R, 2 * D, device="cuda") * 2).half()
    y = fused.geglu(h)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fused.geglu(h)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    a, g = h[:4096].double().chunk(2, dim=-1)
    ref = a * torch.nn.functional.gelu(g)
    err = (y[:4096].double() - ref).abs()
    print(json.dumps({"R": R, "D": D, "us": round(us, 1), "GBps": round(R * D * 6 / us / 1e3, 1), "max_err_over_tol": float((err / (2.0 ** -11 * (1 + ref.abs()))).max())}))
