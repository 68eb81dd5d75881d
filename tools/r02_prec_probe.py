"""Round-2 probe (GPU): measured parity errors of the PRODUCT path per 16-bit type, and GEMM times of the
to_q / to_out projections at the bench shapes (the budget a fused projection has to beat).
Writes gpurun_out/r02_prec_probe.json."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "diffusion-spacetime-attn_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("STA_CONV_FIND", "0")
from oracle import golden_inputs as gi  # noqa: E402
from sta.synth import seeded_fill_  # noqa: E402

G = gi.GOLDEN
res = {"maps": {}, "block_out": {}, "x0": {}, "gemm_us": {}}


def block_maps(name, dtype):
    from ldm.modules.attention import BasicTransformerBlock
    from sta import prompt_state
    g = np.load(os.path.join(G, "block_%s.npz" % name))
    dim, C, heads, K, seed = (int(g[k]) for k in ("dim", "C", "heads", "K", "seed"))
    x, context, local_ctx = gi.block_inputs(dim, C, K, seed, gi.load_uncond())
    blk = BasicTransformerBlock(C, heads, C // heads, context_dim=768, checkpoint=False)
    seeded_fill_(blk, seed)
    blk = blk.to("cuda", dtype)
    blk.keep_maps = True
    prompt_state.begin_prompt([c.cuda() for c in local_ctx], first_timestep=981)
    with torch.no_grad():
        out = blk(x.cuda().to(dtype), context=context.cuda().to(dtype), time=torch.tensor(981),
                  coef=torch.from_numpy(g["coef"]).cuda(), bboxs_curr=[list(c) for c in g["centres"]])
    pix = torch.from_numpy(g["map_pixels"]).cuda()
    got = blk.last_maps[:, :, pix, :].cpu().numpy()
    return float(np.abs(got - g["maps"]).max()), float(np.abs(out.float().cpu().numpy() - g["out"]).max() / np.abs(g["out"]).max())


for name in ("d40", "d80", "d160", "d8k4"):
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        m, o = block_maps(name, dt)
        res["maps"]["%s_%s" % (name, tag)] = m
        res["block_out"]["%s_%s" % (name, tag)] = o


def traj(dtype):
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from ldm.models.diffusion.plms import PLMSSampler
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from sta import prompt_state
    g = np.load(os.path.join(G, "plms_traj.npz"))
    meta = json.load(open(os.path.join(G, "unet_state_dict_keys.json")))
    unet = UNetModel(**meta["cfg"]).eval()
    seeded_fill_(unet, 21)
    for p in unet.parameters():
        p.requires_grad_(False)
    model = LatentDiffusion(unet_config=unet.to("cuda", dtype)).cuda()
    c, local_ctx, x_T = gi.unet_inputs(2, int(g["input_seed"]))
    sampler = PLMSSampler(model, opt_epochs=0, use_graph=False, save_images=False)
    sampler.make_schedule(int(g["S"]), verbose=False)
    time_range = np.flip(sampler.ddim_timesteps)
    W = torch.from_numpy(g["W"]).cuda()
    with torch.no_grad():
        prompt_state.begin_prompt([l.cuda() for l in local_ctx], first_timestep=int(time_range[0]))
        img = sampler._trajectory(x_T.cuda(), c.cuda(), gi.load_uncond().cuda(), float(g["scale"]), time_range, W,
                                  [list(cc) for cc in g["centres"]], 0, graph=False)
    ref = g["x0"]
    err = np.abs(img.float().cpu().numpy() - ref)
    return dict(max_rel=float(err.max() / np.abs(ref).max()), mean_rel=float(err.mean() / np.abs(ref).mean()))


for dt, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
    res["x0"][tag] = traj(dt)


def tm(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for I in (1, 16):
    for N, C in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
        for dt, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
            y = torch.randn(2 * I, N, C, device="cuda", dtype=dt)
            w = torch.randn(C, C, device="cuda", dtype=dt) * 0.05
            b = torch.randn(C, device="cuda", dtype=dt)
            res["gemm_us"]["to_q_I%d_N%d_C%d_%s" % (I, N, C, tag)] = round(tm(lambda: torch.nn.functional.linear(y, w)), 2)
            res["gemm_us"]["to_out_I%d_N%d_C%d_%s" % (I, N, C, tag)] = round(tm(lambda: torch.nn.functional.linear(y, w, b)), 2)

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r02_prec_probe.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
