import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sta import lib, ops
from test_kernel_gpu import _case
for (N, C, heads, K, M) in [(1024, 640, 8, 3, 1), (256, 320, 8, 2, 1), (1024, 640, 8, 3, 2)]:
    q, k, v, mask, coef = _case(N, C, heads, K, torch.bfloat16, seed=9, M=M)
    scale = (C // heads) ** -0.5
    g = torch.Generator().manual_seed(11)
    dout = torch.randn(2, N, C, generator=g).to(torch.bfloat16)
    packed = ops.pack_kv(k.cuda(), v.cuda(), heads)
    dq, dcoef = ops.xattn_backward(q.cuda(), packed, ops.mask_bits(mask).cuda(), coef.cuda(), dout.cuda(), scale)
    d = dq.float().cpu()
    nz = d.abs() > 1e-6
    print(N, C, K, M, "max |dq|", d.abs().max().item(), "nonzero", int(nz.sum()), "nan", int(torch.isnan(d).sum()), "rows", nz.any(-1).any(-1).tolist(), "dcoef", dcoef.tolist())
    if nz.any():
        idx = nz.nonzero()[:10].tolist()
        print("  first:", [(r, p, c, d[r, p, c].item(), bool(mask[:, p].any())) for r, p, c in idx])
