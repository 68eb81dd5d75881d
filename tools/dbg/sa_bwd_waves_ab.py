"""Self-attention backward at SD-v1 level 0 (B = 32 = 16 prompts, N = 4096, d = 40): eight waves per workgroup (default) against four
(STA_OPT_SELFATTN_WAVES = 4), alternated in one process; bit-equality of the gradients; HIP events."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import lib, ops
B, N, C, heads = int(os.environ.get("SA_B", 32)), 4096, 320, 8
scale = ops.LN2 if os.environ.get('SA_PRE', '1') == '1' else (C // heads) ** -0.5
res = {}
for dt in (torch.float16, torch.bfloat16):
    qkv = torch.randn(B, N, 3 * C, device="cuda"); qkv[..., :C] *= 0.228; qkv = qkv.to(dt).requires_grad_(True)
    dout = torch.randn(B, N, C, device="cuda").to(dt)
    out = ops.SelfAttentionQKV.apply(qkv, heads, scale)
    grads = {}
    for rnd in range(3):
        for w in (0, 4):
            lib.set_option(lib.OPT_SELFATTN_WAVES, w)
            g = torch.autograd.grad(out, qkv, dout, retain_graph=True)[0]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                torch.autograd.grad(out, qkv, dout, retain_graph=True)
            e1.record(); torch.cuda.synchronize()
            res.setdefault("%s waves=%s" % (dt, w or 8), []).append(round(e0.elapsed_time(e1) * 100, 1))
            grads[w] = g
    lib.set_option(lib.OPT_SELFATTN_WAVES, 0)
    res["%s bit_equal" % dt] = bool(torch.equal(grads[0], grads[4]))
print(json.dumps(res, indent=1))
