"""Self-attention forward at the bench shapes (64 images x 8 heads): the scaled path against the log2-domain path the
module runs (scale * log2 e folded into W_q, scale = ln 2). Stand-alone or under tools/lib_ab.py."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusion-spacetime-attn_amd"))
from sta import ops  # noqa: E402


def timed(fn, iters=int(os.environ.get("SA_ITERS", "30"))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for B, N, C, h in ((64, 4096, 320, 8), (64, 1024, 640, 8)):
    d = C // h
    qk = torch.randn(B, N, 2 * C, device="cuda", dtype=torch.float16)
    vt = torch.randn(B, C, N, device="cuda", dtype=torch.float16)
    from sta import lib
    a = timed(lambda: ops.self_attention(qk[..., :C], qk[..., C:], vt, h, d ** -0.5))
    b = timed(lambda: ops.self_attention(qk[..., :C], qk[..., C:], vt, h, ops.LN2))
    lib.load().sta_set_option(lib.OPT_SELFATTN_32, 2)          # the 16x16x32-MFMA kernel where the 32x32x16 one is the default (d = 40)
    c = timed(lambda: ops.self_attention(qk[..., :C], qk[..., C:], vt, h, ops.LN2))
    lib.load().sta_set_option(lib.OPT_SELFATTN_32, 0)
    res = {"B": B, "N": N, "d": d, "scaled_us": round(a, 1), "log2_us": round(b, 1), "log2_16x16x32_us": round(c, 1)}
    if d <= 48:
        for waves in (4, 8, 4, 8):          # the two geometries of the d <= 48 kernel, interleaved
            lib.set_option(lib.OPT_SELFATTN_WAVES, waves)
            res.setdefault("log2_%dwaves_us" % waves, []).append(round(timed(lambda: ops.self_attention(qk[..., :C], qk[..., C:], vt, h, ops.LN2)), 1))
        lib.set_option(lib.OPT_SELFATTN_WAVES, 0)
        flop = 4.0 * B * h * N * N * d
        res["tflops_best"] = round(flop / min(res["log2_8waves_us"] + res["log2_4waves_us"]) / 1e6, 1)
    print(json.dumps(res))
