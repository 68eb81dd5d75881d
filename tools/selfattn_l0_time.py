"""The level-0 self-attention launch of the bench (64 x 8 heads, N = 4096, d = 40, fp16, q in log2 units), timed alone: the tool
tools/asm_patch_ab.py runs against every ablation build of csrc/sta_selfattn.hip (flag:-DSTA_SA_ABLATE=bits) and tools/pmc_sa_pipe.sh
profiles. STA_SA_MODE / STA_SA_WAVES select the loop (sta_set_option STA_OPT_SELFATTN_PIPE / _WAVES)."""
import json, os, sys, torch
sys.path.insert(0, "/root/repo/diffusion-spacetime-attn_amd")
from sta import lib, ops
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
B, N, C, h = 64, 4096, 320, 8
dt = torch.bfloat16 if os.environ.get("STA_SA_DTYPE", "fp16") == "bf16" else torch.float16
ops.SELFATTN_OPTIMISTIC = os.environ.get("STA_SA_OPT", "1") != "0"      # 0: the standard loop (running maximum)
g = torch.Generator(device="cuda").manual_seed(0)
qk = torch.randn(B, N, 2 * C, device="cuda", generator=g)
qk[..., :C] *= float(os.environ.get("STA_SA_QSCALE", "1.0"))      # logits: sigma = 6.3 log2 units x this (1.0: row maxima 15+ units above any one block's)
qk = qk.to(dt)
vt = torch.randn(B, C, N, device="cuda", generator=g).to(dt)
run = lambda: ops.self_attention(qk[..., :C], qk[..., C:], vt, h, ops.LN2)
lib.set_option(lib.OPT_SELFATTN_PIPE, int(os.environ.get("STA_SA_MODE", "0")))
lib.set_option(lib.OPT_SELFATTN_WAVES, int(os.environ.get("STA_SA_WAVES", "0")))
t1, t2 = timed(run), timed(run)
st = [(int(w[0]), int(w[1]), int(w[32:].sum())) for w in (f.view(torch.int32).cpu() for f in ops._SA_FLAGS.values())]
print("level-0 self-attention forward %s optimistic=%s qscale=%s: %.1f %.1f us; (sit-out, failures, flagged) = %s" % (dt, ops.SELFATTN_OPTIMISTIC, os.environ.get("STA_SA_QSCALE", "1.0"), t1, t2, st))
