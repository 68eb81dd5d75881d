"""Which part of context 0's backward goes wrong in the failing build? For every bad (pixel, head) of row 0: compare the kernel's dq
with fp64 alternatives in which the -wsum * (VQ.dO1) update of dP is dropped for key tile t (t = 0..4), or for all tiles."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sta import lib, ops
from test_kernel_gpu import _case
N, C, heads, K, dtype = int(os.environ.get("DN", "9216")), 320, 8, 2, torch.float16
d = C // heads
q, k, v, mask, coef = _case(N, C, heads, K, dtype, seed=1)
scale = d ** -0.5
g = torch.Generator().manual_seed(7)
dout = torch.randn(2, N, C, generator=g).to(dtype)
packed = ops.pack_kv(k.cuda(), v.cuda(), heads)
mb = ops.mask_bits(mask).cuda()
lib.set_option(lib.OPT_STAGED_QT, 1)
ref = ops.xattn_backward(q.cuda(), packed, mb, coef.cuda(), dout.cuda(), scale)[0][0].float().cpu()
lib.set_option(lib.OPT_STAGED_QT, 0)
wsum = (mask.double() * coef.double()[:, None]).sum(0)
K0, V0 = k[0].double().view(77, heads, d), v[0].double().view(77, heads, d)
for rep in range(2):
    got = ops.xattn_backward(q.cuda(), packed, mb, coef.cuda(), dout.cuda(), scale)[0][0].float().cpu()
    err = (got - ref).abs().view(N, heads, d).max(-1).values
    bad = (err > 8 * 2.0 ** -11 * ref.abs().max()).nonzero().tolist()
    print("run", rep, "bad (pixel, head) pairs:", len(bad))
    for p, h in bad[:6]:
        qv, g0, g1 = q[0, p].double().view(heads, d)[h], dout[0, p].double().view(heads, d)[h], dout[1, p].double().view(heads, d)[h]
        P = torch.softmax(scale * (K0[:, h] @ qv), 0)
        dpa, dpb = V0[:, h] @ g0, V0[:, h] @ g1
        def dq_of(dp):
            return scale * ((P * (dp - (P * dp).sum())) @ K0[:, h])
        full = dq_of(dpa - wsum[p] * dpb)
        gk = got.view(N, heads, d)[p, h].double()
        res = {"full": (gk - full).abs().max().item(), "none": (gk - dq_of(dpa)).abs().max().item()}
        for t in range(5):
            dp = dpa - wsum[p] * dpb
            dp[16 * t:16 * t + 16] = dpa[16 * t:16 * t + 16]
            res["drop%d" % t] = (gk - dq_of(dp)).abs().max().item()
        for gg in range(4):       # the four keys 16t + 4g + r of every tile held by lane row g
            dp = dpa - wsum[p] * dpb
            idx = [16 * t + 4 * gg + r for t in range(5) for r in range(4) if 16 * t + 4 * gg + r < 77]
            dp[idx] = dpa[idx]
            res["droprow%d" % gg] = (gk - dq_of(dp)).abs().max().item()
        def alt(qq, gg0, gg1):
            P2 = torch.softmax(scale * (K0[:, h] @ qq), 0)
            dp2 = V0[:, h] @ gg0 - wsum[p] * (V0[:, h] @ gg1)
            return scale * ((P2 * (dp2 - (P2 * dp2).sum())) @ K0[:, h])
        z = lambda x, a, b: torch.cat([x[:a], torch.zeros(b - a, dtype=x.dtype), x[b:]])
        q1v = q[1, p].double().view(heads, d)[h]
        for name, (qq, gg0, gg1) in {"g0hi0": (qv, z(g0, 32, 40), g1), "g0lo0": (qv, z(g0, 0, 32), g1), "q0hi0": (z(qv, 32, 40), g0, g1), "q0lo0": (z(qv, 0, 32), g0, g1),
                                     "g1hi0": (qv, g0, z(g1, 32, 40)), "g1lo0": (qv, g0, z(g1, 0, 32)), "q=q1": (q1v, g0, g1), "g0=g1": (qv, g1, g1)}.items():
            res[name] = (gk - alt(qq, gg0, gg1)).abs().max().item()
        for t in range(5):      # S^T of key tile t without its second k-step (head dims 32..39) / first k-step
            for nm, (a, b) in {"st%d_nohi" % t: (32, 40), "st%d_nolo" % t: (0, 32)}.items():
                sc_ = scale * (K0[:, h] @ qv)
                sc_[16 * t:16 * t + 16] = scale * (K0[16 * t:16 * t + 16, h] @ z(qv, a, b))
                P2 = torch.softmax(sc_, 0)
                dp2 = dpa - wsum[p] * dpb
                res[nm] = (gk - scale * ((P2 * (dp2 - (P2 * dp2).sum())) @ K0[:, h])).abs().max().item()
                dp2 = dpa - wsum[p] * dpb
                dp2[16 * t:16 * t + 16] = V0[16 * t:16 * t + 16, h] @ z(g0, a, b) - wsum[p] * dpb[16 * t:16 * t + 16]
                res["dp%d_%s" % (t, nm[-4:])] = (gk - dq_of(dp2)).abs().max().item()
        r = gk - full
        Kh = K0[:, h]
        c = (Kh @ r) / (Kh * Kh).sum(1)
        rem = (r[None, :] - c[:, None] * Kh).norm(dim=1) / r.norm()
        kb = int(rem.argmin())
        dp_true = dpa - wsum[p] * dpb
        dS_true = P * (dp_true - (P * dp_true).sum())
        # least squares over the 4 keys of one accumulator register group (tile t, lane row g): keys 16 t + 4 g + 0..3
        bestg = None
        for t in range(5):
            for gg in range(4):
                idx = [16 * t + 4 * gg + rr for rr in range(4) if 16 * t + 4 * gg + rr < 77]
                if not idx:
                    continue
                A = Kh[idx].T                                   # [d, 4]
                sol = torch.linalg.lstsq(A, r[:, None]).solution[:, 0]
                rr_ = (r - A @ sol).norm() / r.norm()
                if bestg is None or rr_ < bestg[0]:
                    bestg = (float(rr_), t, gg, (sol / scale).tolist(), dS_true[idx].tolist(), P[idx].tolist())
        print("     single key: k=%d unexplained %.2f (dS err %.3g, true dS %.3g, P %.3g) | group tile %d row %d unexplained %.2f dS err %s true %s P %s" % (
            kb, rem[kb], c[kb] / scale, dS_true[kb], P[kb], bestg[1], bestg[2], bestg[0], ["%.3g" % x for x in bestg[3]], ["%.3g" % x for x in bestg[4]], ["%.3g" % x for x in bestg[5]]))
        k30 = kb
        # which other value would key kb's (dp - delta) have to be replaced by to explain the error? candidates: the same quantity of the
        # other keys of the lane row (r = 0..3 of every tile: 16 t + 4 g + r)
        delta_true = (P * dp_true).sum()
        need = (dS_true[kb] + c[kb] / scale) / P[kb]            # the (dp - delta) the kernel must have used, if P is right
        gg = (kb % 16) // 4
        cands = {"t%d r%d" % (t, rr): float(dp_true[16 * t + 4 * gg + rr] - delta_true) for t in range(5) for rr in range(4) if 16 * t + 4 * gg + rr < 77}
        bestc = min(cands, key=lambda n_: abs(cands[n_] - need))
        needP = (dS_true[kb] + c[kb] / scale) / (dp_true[kb] - delta_true)     # or the P it must have used, if dp - delta is right
        candP = {"t%d r%d" % (t, rr): float(P[16 * t + 4 * gg + rr]) for t in range(5) for rr in range(4) if 16 * t + 4 * gg + rr < 77}
        bestP = min(candP, key=lambda n_: abs(candP[n_] - needP))
        vg_lo, vg_hi = float(V0[kb, h, :32] @ g0[:32]), float(V0[kb, h, 32:] @ g0[32:])
        kq_lo, kq_hi = float(scale * (Kh[kb, :32] @ qv[:32])), float(scale * (Kh[kb, 32:] @ qv[32:]))
        import math as _m
        print("     key %d: if dp is off: delta_dp = %+.4f  [V.g0 dims 0..31 = %+.4f, dims 32..39 = %+.4f] | if P is off: ln(P'/P) = %+.4f  [scale K.q dims 0..31 = %+.4f, dims 32..39 = %+.4f]" % (
            kb, need - float(dp_true[kb] - delta_true), vg_lo, vg_hi, _m.log(max(float(needP / P[kb]), 1e-9)), kq_lo, kq_hi))
        tv = torch.tensor([float(dS_true[kb] * scale), float(dS_true[kb] * scale + c[kb])], dtype=torch.float32)
        hv = tv.to(torch.float16).view(torch.int16).tolist()
        print("     key %d: scaled dS true %.6f (fp16 0x%04x)  used %.6f (fp16 0x%04x)  xor 0x%04x  diff %.6f = 2^%.2f" % (kb, tv[0], hv[0] & 0xffff, tv[1], hv[1] & 0xffff, (hv[0] ^ hv[1]) & 0xffff, tv[1] - tv[0], _m.log2(abs(float(tv[1] - tv[0])) + 1e-30)))
        print("     key %d = tile %d row %d r %d: needs (dp - delta) = %.4f (true %.4f); closest other key of the lane row: %s = %.4f | or P = %.5f (true %.5f); closest: %s = %.5f" % (
            kb, kb // 16, gg, kb % 4, need, dp_true[kb] - delta_true, bestc, cands[bestc], needP, P[kb], bestP, candP[bestP]))
        def dS_with(P_, dp_):
            return P_ * (dp_ - (P_ * dp_).sum())
        hyp = {}
        dpx = dp_true.clone(); dpx[k30] = V0[k30, h, :32] @ g0[:32] - wsum[p] * dpb[k30]; hyp["dp: first k-step of VQ.dO0 only"] = dS_with(P, dpx)[k30] - dS_true[k30]
        dpx = dp_true.clone(); dpx[k30] = V0[k30, h, 32:] @ g0[32:] - wsum[p] * dpb[k30]; hyp["dp: second k-step only"] = dS_with(P, dpx)[k30] - dS_true[k30]
        dpx = dp_true.clone(); dpx[k30] = dpa[k30] - wsum[p] * (V0[k30, h, :32] @ g1[:32]); hyp["ab: first k-step only"] = dS_with(P, dpx)[k30] - dS_true[k30]
        dpx = dp_true.clone(); dpx[k30] = dpa[k30]; hyp["no wsum update"] = dS_with(P, dpx)[k30] - dS_true[k30]
        dpx = dp_true.clone(); dpx[k30] = 0; hyp["dp = 0"] = dS_with(P, dpx)[k30] - dS_true[k30]
        sc_ = scale * (Kh @ qv); sc2 = sc_.clone(); sc2[k30] = scale * (Kh[k30, :32] @ qv[:32]); hyp["S: first k-step only"] = dS_with(torch.softmax(sc2, 0), dp_true)[k30] - dS_true[k30]
        sc2 = sc_.clone(); sc2[k30] = scale * (Kh[k30, 32:] @ qv[32:]); hyp["S: second k-step only"] = dS_with(torch.softmax(sc2, 0), dp_true)[k30] - dS_true[k30]
        # what dp[k30] would explain the fitted error exactly?
        need_dp = dp_true[k30] + (c[k30] / scale) / (P[k30] * (1 - P[k30]))
        print("     fitted dS err %.4g; hypotheses: %s | dp_true %.3f, dp that would explain it %.3f (dpa %.3f, dpb %.3f, wsum*dpb %.3f)" % (
            c[k30] / scale, "; ".join("%s %.4g" % kv for kv in hyp.items()), dp_true[k30], need_dp, dpa[k30], dpb[k30], wsum[p] * dpb[k30]))
        best = min(res, key=res.get)
        print("  px %5d (tile %2d wave %d c16 %2d) head %d wsum %.2f: best %-8s %.2e | full %.2e none %.2e" % (p, p // 128, (p % 128) // 16, p % 16, h, wsum[p], best, res[best], res["full"], res["none"]))
lib.set_option(lib.OPT_STAGED_QT, 0)
