"""Micro-benchmark of the flash-style self-attention kernel (attn1) at the two UNet levels it serves."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for (N, C) in [(4096, 320), (1024, 640)]:
    H = 8
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, N, C, generator=g).bfloat16().cuda()
    k = torch.randn(B, N, C, generator=g).bfloat16().cuda()
    vt = torch.randn(B, C, N, generator=g).bfloat16().cuda()
    sc = (C // H) ** -0.5
    for _ in range(3):
        ops.self_attention(q, k, vt, H, sc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.self_attention(q, k, vt, H, sc)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("B=%d N=%d C=%d: %.1f us  %.0f TFLOP/s" % (B, N, C, us, 4.0 * B * N * N * C / us / 1e6))
