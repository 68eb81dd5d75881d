"""Probes for the p3 kernel: which stage breaks for the cond row (debug aid)."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import lib, ops  # noqa: E402

dev, dtype = "cuda", torch.float16
N, C, heads, M, I, K = 256, 320, 8, 77, 1, 0
g = torch.Generator().manual_seed(1)
wq = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dtype).to(dev)
scale = (C // heads) ** -0.5


def p3(y, k, v):
    lib.set_option(lib.OPT_PROJ_PAIR, 3)
    out = ops.xattn_forward_proj(y, ops.pack_wq(wq, heads), ops.pack_kv_proj(k, v, heads, n_img=I), None, None, scale)
    lib.set_option(lib.OPT_PROJ_PAIR, 0)
    torch.cuda.synchronize()
    return out.float()


def ref(y, k, v):
    q = torch.nn.functional.linear(y, wq)
    r, _ = ops.xattn_forward(q, ops.pack_kv(k, v, heads, n_img=I), None, None, scale)
    torch.cuda.synchronize()
    return r.float()


y = torch.randn(2, N, C, generator=g).to(dtype).to(dev)
k = (torch.randn(2, M, C, generator=g) * 0.7).to(dtype).to(dev)
v = torch.randn(2, M, C, generator=g).to(dtype).to(dev)
o, r = p3(y, k, v), ref(y, k, v)
print("base: row0 %.4g row1 %.4g" % ((o[0] - r[0]).abs().max(), (o[1] - r[1]).abs().max()))
# P1: identical contexts and rows
y1 = torch.stack([y[0], y[0]]); k1 = torch.stack([k[0], k[0]]); v1 = torch.stack([v[0], v[0]])
o = p3(y1, k1, v1)
print("P1 same ctx, same row: |row1 - row0| %.4g" % (o[1] - o[0]).abs().max())
# P1b: identical contexts, different rows
o, r = p3(y, k1, v1), ref(y, k1, v1)
print("P1b same ctx, rows differ: row0 %.4g row1 %.4g" % ((o[0] - r[0]).abs().max(), (o[1] - r[1]).abs().max()))
# P1c: different contexts, same rows
o, r = p3(y1, k, v), ref(y1, k, v)
print("P1c ctx differ, same rows: row0 %.4g row1 %.4g" % ((o[0] - r[0]).abs().max(), (o[1] - r[1]).abs().max()))
# P2: V = 1
o = p3(y, k, torch.ones_like(v))
print("P2 V=1: row0 max|o-1| %.4g row1 %.4g; row1 sample %s" % ((o[0] - 1).abs().max(), (o[1] - 1).abs().max(), o[1, 0, :6].tolist()))
# P3: K = 0 -> mean of V
o = p3(y, torch.zeros_like(k), v)
m = v.float().mean(dim=1)                     # [2, C]
print("P3 K=0: row0 %.4g row1 %.4g" % ((o[0] - m[0]).abs().max(), (o[1] - m[1]).abs().max()))
# P4: swap contexts: does the error follow the context or the row?
ks, vs = k.flip(0), v.flip(0)
o, r = p3(y, ks, vs), ref(y, ks, vs)
print("P4 swapped ctx: row0 %.4g row1 %.4g" % ((o[0] - r[0]).abs().max(), (o[1] - r[1]).abs().max()))
# where in the row are the errors (per head dim)
o, r = p3(y, k, v), ref(y, k, v)
d = (o[1] - r[1]).abs().view(N, heads, 40).amax(dim=0)
print("row1 err per head (max over px) by dim:")
for h in range(heads):
    print("  h%d " % h + " ".join("%5.1f" % x for x in d[h].tolist()))
