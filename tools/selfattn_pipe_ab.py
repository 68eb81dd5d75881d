"""Self-attention forward at the level-0 bench shape (64 images x 8 heads, N = 4096, d = 40, q in log2 units): the software-pipelined
loop (default) against the plain loop (STA_OPT_SELFATTN_PIPE = 2), interleaved on the same box; results of the two compared too."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusion-spacetime-attn_amd"))
from sta import lib, ops  # noqa: E402


def timed(fn, iters=int(os.environ.get("SA_ITERS", "30"))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for dtype in (torch.float16, torch.bfloat16):
    for B, N in ((64, 4096), (8, 4096), (4, 9216)):
        C, h = 320, 8
        d = C // h
        g = torch.Generator(device="cuda").manual_seed(0)
        qk = torch.randn(B, N, 2 * C, device="cuda", generator=g).to(dtype)
        vt = torch.randn(B, C, N, device="cuda", generator=g).to(dtype)
        run = lambda: ops.self_attention(qk[..., :C], qk[..., C:], vt, h, ops.LN2)
        res = {"dtype": str(dtype)[6:], "B": B, "N": N, "d": d}
        outs = {}
        names = {(4, 0): "pipelined_4x2", (3, 0): "pipelined_4x3", (8, 0): "pipelined_8x2", (2, 0): "plain_4x2"}
        for rnd in range(2):
            for (mode, waves), name in names.items():
                lib.set_option(lib.OPT_SELFATTN_PIPE, mode)
                lib.set_option(lib.OPT_SELFATTN_WAVES, waves)
                outs[name] = run().float()
                res.setdefault(name + "_us", []).append(round(timed(run), 1))
        lib.set_option(lib.OPT_SELFATTN_PIPE, 0)
        lib.set_option(lib.OPT_SELFATTN_WAVES, 0)
        res["max_abs_diff"] = max((outs[n] - outs["plain_4x2"]).abs().max().item() for n in outs)
        flop = 4.0 * B * h * N * N * d
        for n in names.values():
            res["tflops_" + n] = round(flop / min(res[n + "_us"]) / 1e6, 1)
        print(json.dumps(res), flush=True)
