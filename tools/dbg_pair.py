"""Every instantiation of the head-pair kernel (row-major / query-fragment / out-fragment, fp16 / bf16) and the unfused staged kernel
against the CPU ORACLE on one seeded case per size — the check that localises a code-generation hazard to an instantiation."""
import sys, math, torch
sys.path.insert(0, "/root/repo/diffusion-spacetime-attn_amd"); sys.path.insert(0, "/root/repo")
from oracle import xattn_oracle as orc
from sta import lib, ops
for dtype in (torch.float16, torch.bfloat16):
  for (N, I) in ((256, 1), (4096, 2)):
    C, heads, K, M = 320, 8, 2, 77
    g = torch.Generator().manual_seed(1)
    y = torch.randn(2 * I, N, C, generator=g).to(dtype)
    wq = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dtype)
    k = torch.randn(I * (K + 2), M, C, generator=g).to(dtype)
    v = torch.randn(I * (K + 2), M, C, generator=g).to(dtype)
    centres = [(0.3, 0.4), (0.7, 0.6)]
    mb = ops.disc_mask_bits(centres, int(N ** 0.5)).repeat(I, 1)
    mk = orc.disc_masks([list(c) for c in centres], int(N ** 0.5)).reshape(K, N)
    coef = torch.full((I, K), 2.5)
    q16 = (y.double() @ wq.double().t()).to(dtype)
    ref = torch.cat([orc.fused_xattn(q16[2 * i:2 * i + 2].double(), k[4 * i:4 * i + 4].double(), v[4 * i:4 * i + 4].double(), mk, coef[i].double(), heads, 40 ** -0.5)
                     for i in range(I)]).float()
    yd, wqd, kd, vd, mbd, cd = y.cuda(), wq.cuda(), k.cuda(), v.cuda(), mb.cuda(), coef.cuda()
    lib.set_option(lib.OPT_PROJ_PAIR, 1)
    wqf, kvp = ops.pack_wq(wqd, heads), ops.pack_kv_proj(kd, vd, heads, n_img=I)
    outs = {"rm": ops.xattn_forward_proj(yd, wqf, kvp, mbd, cd, 40 ** -0.5),
            "qf": ops.xattn_forward_proj(ops.to_qfrag(yd), wqf, kvp, mbd, cd, 40 ** -0.5, qfrag=True),
            **({"of": ops.from_ofrag(ops.xattn_forward_proj(ops.to_qfrag(yd), wqf, kvp, mbd, cd, 40 ** -0.5, qfrag=True, ofrag=True))}
               if ops.proj_ofrag_supported(C, heads, dtype) else {})}
    lib.set_option(lib.OPT_PROJ_PAIR, 0)
    outs["staged(unfused)"] = ops.xattn_forward(torch.nn.functional.linear(yd, wqd), ops.pack_kv(kd, vd, heads, n_img=I), mbd, cd, 40 ** -0.5)[0]
    torch.cuda.synchronize()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for name, o in outs.items():
        o = o.float().cpu()
        bad = ((o - ref).abs() > 4 * eps * (1 + ref.abs()))
        msg = "%s N=%d I=%d %-16s bad frac %.5f" % (str(dtype)[6:], N, I, name, bad.float().mean().item())
        if bad.any():
            idx = bad.nonzero()
            px = idx[:, 1].unique()
            msg += " rows %s; %d px; chans %d (first %s); in-disc frac %.2f" % (
                idx[:, 0].unique().tolist(), px.numel(), idx[:, 2].unique().numel(), idx[:, 2].unique()[:12].tolist(), (mb[0][px] != 0).float().mean().item())
        print(msg, flush=True)
