"""Where does the backward differ from the fp64 oracle? usage: python tools/dbg/bwd_err.py N C heads K [I]"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd")); sys.path.insert(0, ROOT)
from sta import lib, ops
from oracle import xattn_oracle as orc
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_kernel_gpu import _case
N, C, heads, K = map(int, sys.argv[1:5])
for dtype in (torch.bfloat16, torch.float16):
    q, k, v, mask, coef = _case(N, C, heads, K, dtype, seed=1)
    scale = (C // heads) ** -0.5
    g = torch.Generator().manual_seed(7)
    dout = torch.randn(2, N, C, generator=g).to(dtype)
    qd = q.double().requires_grad_(True); cd = coef.double().requires_grad_(True)
    orc.fused_xattn(qd, k.double(), v.double(), mask, cd, heads, scale).backward(dout.double())
    packed = ops.pack_kv(k.cuda(), v.cuda(), heads)
    mb = ops.mask_bits(mask).cuda()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for opt in (1, 2):
        lib.set_option(lib.OPT_BWD_KERNEL, opt)
        dq, dcoef = ops.xattn_backward(q.cuda(), packed, mb, coef.cuda(), dout.cuda(), scale)
        torch.cuda.synchronize()
        err = (dq.float().cpu().double() - qd.grad).abs()
        gs = qd.grad.abs().max().item()
        anyd = mask.any(0) if K else torch.zeros(N, dtype=torch.bool)
        print(dtype, "kernel", opt, "tol %.4g" % (6 * eps * gs), "row0 in/out disc %.4g %.4g  row1 in/out %.4g %.4g" % (
            err[0][anyd].max() if anyd.any() else 0, err[0][~anyd].max(), err[1][anyd].max() if anyd.any() else 0, err[1][~anyd].max()))
        r, p, c = [int(x) for x in (err == err.max()).nonzero()[0]]
        print("   max at row %d px %d ch %d (head %d): got %.5f want %.5f ; mask bits %s; nbad %d" % (r, p, c, c // (C // heads), dq[r, p, c].item(), qd.grad[r, p, c].item(), mask[:, p].tolist(), int((err > 6 * eps * gs).sum())))
        if opt == 1:
            bad = (err > 3 * eps * gs).nonzero()
            grp_any = mask.any(0).view(-1, 16).any(1)
            import collections
            cnt = collections.Counter()
            for r, p, c in bad.tolist()[:4000]:
                cnt[(r, "grp_touches_disc=%d" % int(grp_any[p // 16]), "px_in_disc=%d" % int(mask[:, p].any()), "head=%d" % (c // (C // heads)), "dim=%d" % (c % (C // heads)))] += 1
            for kk, vv in sorted(cnt.items(), key=lambda x: -x[1])[:25]:
                print("     ", kk, vv)
            pxs = sorted(set(p for r, p, c in bad.tolist()))
            print("      bad pixels:", len(pxs), pxs[:40])
        if K:
            print("   dcoef", dcoef.cpu().tolist(), "want", cd.grad.tolist())
    lib.set_option(lib.OPT_BWD_KERNEL, 0)
