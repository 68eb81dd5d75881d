"""Differentiable self-attention (sta.ops.SelfAttentionQKV: HIP forward with lse + HIP backward) against PyTorch SDPA
autograd at the attn1 shapes of the tracked epochs (one prompt = CFG batch 2). HIP events on the current stream."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import ops  # noqa: E402


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dtype = torch.float16 if "--bf16" not in sys.argv else torch.bfloat16
    for B, N, C, heads in ((2, 4096, 320, 8), (2, 1024, 640, 8), (2, 9216, 320, 8), (8, 4096, 320, 8)):
        d, scale = C // heads, (C // heads) ** -0.5
        qkv = torch.randn(B, N, 3 * C, device="cuda").to(dtype).requires_grad_(True)
        dout = torch.randn(B, N, C, device="cuda").to(dtype)
        out = ops.SelfAttentionQKV.apply(qkv, heads, scale)
        t_fwd = timed(lambda: ops.SelfAttentionQKV.apply(qkv, heads, scale))
        t_bwd = timed(lambda: torch.autograd.grad(out, qkv, dout, retain_graph=True))

        def sdpa():
            q, k, v = (qkv[..., i * C:(i + 1) * C].view(B, N, heads, d).transpose(1, 2) for i in range(3))
            return F.scaled_dot_product_attention(q, k, v, scale=scale).transpose(1, 2).reshape(B, N, C)
        o2 = sdpa()
        t_fwd2 = timed(sdpa)
        t_bwd2 = timed(lambda: torch.autograd.grad(o2, qkv, dout, retain_graph=True))
        gflop = 4.0 * N * N * d * heads * B / 1e9
        print(json.dumps({"B": B, "N": N, "C": C, "d": d, "dtype": str(dtype), "hip_fwd_us": round(t_fwd, 1), "hip_bwd_us": round(t_bwd, 1),
                          "sdpa_fwd_us": round(t_fwd2, 1), "sdpa_bwd_us": round(t_bwd2, 1), "fwd_gflop": round(gflop, 1),
                          "hip_bwd_tflops": round(2.5 * gflop / t_bwd * 1e3, 1), "sdpa_bwd_tflops": round(2.5 * gflop / t_bwd2 * 1e3, 1)}))


if __name__ == "__main__":
    main()
