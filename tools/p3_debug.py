"""Error map of the second-generation head-pair kernel (OPT_PROJ_PAIR = 3) against the unfused GPU path:
max |diff| per (batch row, head, 16-pixel wave tile). Debug aid for tools/ only."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import lib, ops  # noqa: E402


def run(N, C, heads, K, I, dtype, ring=0):
    dev, M = "cuda", 77
    g = torch.Generator().manual_seed(N + C + K)
    wq = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dtype).to(dev)
    y = torch.randn(2 * I, N, C, generator=g).to(dtype).to(dev)
    k = (torch.randn(I * (K + 2), M, C, generator=g) * 0.7).to(dtype).to(dev)
    v = torch.randn(I * (K + 2), M, C, generator=g).to(dtype).to(dev)
    dim = int(math.isqrt(N))
    cent = [(0.30, 0.40), (0.70, 0.60)][:K]
    mask = ops.disc_mask_bits(cent, dim).to(dev).repeat(I, 1) if K else None
    coef = torch.full((I, K), 2.5, device=dev) if K else None
    scale = (C // heads) ** -0.5
    lib.set_option(lib.OPT_PROJ_PAIR, 3)
    lib.set_option(lib.OPT_PROJ_RING, ring)
    out = ops.xattn_forward_proj(y, ops.pack_wq(wq, heads), ops.pack_kv_proj(k, v, heads, n_img=I), mask, coef, scale)
    lib.set_option(lib.OPT_PROJ_PAIR, 0)
    lib.set_option(lib.OPT_PROJ_RING, 0)
    q = torch.nn.functional.linear(y, wq)
    ref, _ = ops.xattn_forward(q, ops.pack_kv(k, v, heads, n_img=I), mask, coef, scale)
    torch.cuda.synchronize()
    d = (out.float() - ref.float()).abs()                      # [2I, N, C]
    d = d.view(I, 2, N // 16, 16, heads, C // heads).amax(dim=(3, 5))      # [I, 2, tiles16, heads]
    print("N=%d C=%d K=%d I=%d %s ring=%d: max diff %.4g, nan %d" % (N, C, K, I, dtype, ring, d.max().item(), int(torch.isnan(out).sum())))
    bad = d > 0.05
    if bad.any():
        for i in range(I):
            for r in range(2):
                rows = ["".join("X" if bad[i, r, t, h] else "." for t in range(min(N // 16, 64))) for h in range(heads)]
                print(" img %d row %d (one line per head, one char per 16 px):" % (i, r))
                for h, line in enumerate(rows):
                    print("   h%d %s" % (h, line))


for dt in (torch.float16, torch.bfloat16):
    for ring in (5, 0):
        run(256, 320, 8, 0, 1, dt, ring)
        run(256, 320, 8, 2, 1, dt, ring)
        run(1024, 320, 8, 2, 2, dt, ring)
