import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "diffusion-spacetime-attn_amd"))
from oracle import golden_inputs as gi
from tests.test_modules_gpu import _golden_unet
from ldm.models.diffusion.ddpm import LatentDiffusion
from ldm.models.diffusion.plms import PLMSSampler
c, local_ctx, x_T = gi.unet_inputs(2, 41)
res = {}
unet = _golden_unet(torch.float16)
model = LatentDiffusion(unet_config=unet).cuda()
for mode in ("eager", "graph", "eager2"):
    sampler = PLMSSampler(model, opt_epochs=0, use_graph=(mode == "graph"), save_images=False)
    for rep in range(2):
        sampler.sample(S=10, conditioning=c.cuda() * (1 + rep), batch_size=1, shape=[4, 32, 32], verbose=False,
                       unconditional_guidance_scale=7.5, unconditional_conditioning=gi.load_uncond().cuda(), eta=0.0,
                       x_T=x_T.cuda(), text_index=0, curr_text="x", bboxs_curr=[[0.3, 0.4], [0.7, 0.6 - 0.1 * rep]], seed=1,
                       prompt_idx=0, object_names=["a", "b"], local_conditionings=[l.cuda() for l in local_ctx])
        res[(mode, rep)] = sampler.last_result["x0"].clone()
for rep in range(2):
    a, b, e2 = res[("eager", rep)], res[("graph", rep)], res[("eager2", rep)]
    print(rep, "eager-vs-graph max diff", (a - b).abs().max().item(), "eager-vs-eager2", (a - e2).abs().max().item(), "scale", a.abs().max().item())
