"""GPU time vs wall time of ONE CFG UNet call forward + backward (w.r.t. the latent and the blend weights) in the
autograd (weight-optimisation) path: is the tracked epoch launch-bound or GPU-bound, and which ops dominate.

usage: python tools/op_profile_bwd.py [images]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
for k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + k, "0")
from sta import prompt_state  # noqa: E402
from sta.pipeline import DEFAULT_CENTRES, build_sd_v1, conditionings, set_recompute, use_shipped_miopen_db  # noqa: E402

use_shipped_miopen_db(0)

if os.environ.get("STA_FA_LIB"):
    torch.backends.cuda.preferred_rocm_fa_library(os.environ["STA_FA_LIB"])
    print("rocm fa library:", torch.backends.cuda.preferred_rocm_fa_library())
I = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
dev, dt, K = torch.device("cuda", 0), (torch.float16 if os.environ.get('DT', 'fp16') == 'fp16' else torch.bfloat16), 2
model = build_sd_v1(dev, dt, with_vae=False, init_weights=True, seed=0, use_checkpoint=True)
set_recompute(model, os.environ.get("POLICY", "none"))       # POLICY=call: NHWC trunk + differentiable fused glue ops
from sta import fused  # noqa: E402
uc, c, local_c = conditionings(model, "a photo of a cat and a dog", ["cat", "dog"], dt)
pair = lambda u, v: torch.stack([u, v], dim=1).reshape(2 * I, *u.shape[1:])
c_in = pair(uc.expand(I, -1, -1), c.expand(I, -1, -1)).contiguous()
t_in = torch.full((2 * I,), 981, device=dev, dtype=torch.long)
centres = [list(cc) for cc in DEFAULT_CENTRES[:K]]
boxes = [centres] * I if I > 1 else centres
prompt_state.begin_prompt([local_c] * I if I > 1 else local_c, first_timestep=981)


def call():
    x = torch.randn(I, 4, 64, 64, device=dev, requires_grad=True)
    coef = (torch.full((I, K), 2.5, device=dev) if I > 1 else torch.full((K,), 2.5, device=dev)).requires_grad_(True)
    with fused.tracked(os.environ.get("POLICY") == "call"):
        out = model.apply_model_extra(pair(x, x), 0, t_in, c_in, coef=coef, bboxs_curr=boxes)
        out.float().square().mean().backward()
    return x.grad, coef.grad


for _ in range(3):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    call()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=bool(os.environ.get('STACK'))) as prof:
    for _ in range(3):
        call()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=50))
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=40, max_shapes_column_width=70))
if os.environ.get("STACK"):
    rows = [e for e in prof.key_averages(group_by_stack_n=8) if e.key in ("aten::copy_", "aten::add_", "aten::add", "aten::cat", "aten::mul")]
    rows.sort(key=lambda e: -e.self_device_time_total)
    for e in rows[:40]:
        print("%-12s %8.2f ms %4d calls" % (e.key, e.self_device_time_total / 1e3, e.count))
        for fr in e.stack[:8]:
            if "site-packages/torch" not in fr and "dist-packages/torch" not in fr:
                print("      ", fr[-150:])
print("wall per fwd+bwd call (un-profiled): %.1f ms" % (wall * 1e3))
