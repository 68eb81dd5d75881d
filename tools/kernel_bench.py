"""Micro-benchmark of the fused cross-attention kernels at the four SD-v1 UNet level shapes.

Per-launch time by HIP events on the launch stream (torch's current stream); algorithmic flops and
bytes per SURVEY.md §8(d): F = 4*M*C*N*(K+2), Bt = 8*N*C + 4*(K+2)*M*C + K*N  (bf16).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import ops  # noqa: E402

LEVELS = {512: [(4096, 320), (1024, 640), (256, 1280), (64, 1280)],
          768: [(9216, 320), (2304, 640), (576, 1280), (144, 1280)]}
CENTRES = [(0.30, 0.40), (0.70, 0.60), (0.5, 0.2), (0.25, 0.75)]


def bench_level(N, C, K, heads=8, M=77, iters=200, dtype=torch.bfloat16, bwd=False, imgs=1):
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)
    q = torch.randn(2 * imgs, N, C, generator=g).to(dtype).to(dev)
    k = (torch.randn(imgs * (K + 2), M, C, generator=g) * 0.78).to(dtype).to(dev)
    v = torch.randn(imgs * (K + 2), M, C, generator=g).to(dtype).to(dev)
    dim = int(N ** 0.5)
    mask = ops.disc_mask_bits(CENTRES[:K], dim).to(dev).repeat(imgs, 1)
    coef = torch.full((imgs, K), 5.0 / max(K, 1), device=dev)
    packed = ops.pack_kv(k, v, heads, n_img=imgs)
    scale = (C // heads) ** -0.5
    dout = torch.randn(2 * imgs, N, C, generator=g).to(dtype).to(dev)
    fn = (lambda: ops.xattn_backward(q, packed, mask, coef, dout, scale)) if bwd else \
         (lambda: ops.xattn_forward(q, packed, mask, coef, scale))
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = imgs * 4.0 * M * C * N * (K + 2) * (2 if bwd else 1)
    byts = imgs * ((12.0 if bwd else 8.0) * N * C + 4.0 * (K + 2) * M * C + K * N)
    return {"N": N, "C": C, "K": K, "imgs": imgs, "bwd": bwd, "us": round(us, 2), "TFLOPs": round(flops / us / 1e6, 1),
            "GBps": round(byts / us / 1e3, 1)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--K", type=int, default=2)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--imgs", type=int, default=1)
    ap.add_argument("--level", type=int, default=-1, help="only this level (0..3) of the resolution")
    ap.add_argument("--kernel", type=int, default=0, help="sta_set_option(STA_OPT_FWD_KERNEL): 0 auto, 1 LDS-resident, 2 split / one context at a time")
    ap.add_argument("--opt", action="append", default=[], help="sta_set_option override KEY=VALUE (KEY = index in include/sta_xattn.h), repeatable")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16")
    a = ap.parse_args()
    from sta import lib
    if a.kernel:
        lib.set_option(lib.OPT_FWD_KERNEL, a.kernel)
    for kv in a.opt:
        k_, v_ = kv.split("=")
        lib.set_option(int(k_), int(v_))
    for N, C in (LEVELS[a.res] if a.level < 0 else [LEVELS[a.res][a.level]]):
        print(json.dumps(bench_level(N, C, a.K, iters=a.iters, bwd=a.bwd, imgs=a.imgs, dtype=torch.float16 if a.dtype == 'fp16' else torch.bfloat16)))
