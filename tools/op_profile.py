"""Operator-level (aten / custom op) GPU time of ONE CFG UNet call at I images per call (torch.profiler).

usage: python tools/op_profile.py [images] [--shapes]   -> table sorted by device time
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
for k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + k, "0")
from sta import prompt_state  # noqa: E402
from sta.pipeline import DEFAULT_CENTRES, build_sd_v1, conditionings  # noqa: E402

if os.environ.get("STA_CUDNN_BENCHMARK") == "1":
    torch.backends.cudnn.benchmark = True
I = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
shapes = "--shapes" in sys.argv
dev, dt, K = torch.device("cuda", 0), torch.bfloat16, 2
model = build_sd_v1(dev, dt, with_vae=False, init_weights=True, seed=0, channels_last="--channels-last" in sys.argv)
uc, c, local_c = conditionings(model, "a photo of a cat and a dog", ["cat", "dog"], dt)
pair = lambda u, v: torch.stack([u, v], dim=1).reshape(2 * I, *u.shape[1:])
c_in = pair(uc.expand(I, -1, -1), c.expand(I, -1, -1)).contiguous()
x_in = torch.randn(2 * I, 4, 64, 64, device=dev)
t_in = torch.full((2 * I,), 981, device=dev, dtype=torch.long)
coef = torch.full((I, K), 2.5, device=dev) if I > 1 else torch.full((K,), 2.5, device=dev)
centres = [list(cc) for cc in DEFAULT_CENTRES[:K]]
boxes = [centres] * I if I > 1 else centres
prompt_state.begin_prompt([local_c] * I if I > 1 else local_c, first_timestep=981)


def call():
    with torch.no_grad():
        return model.apply_model_extra(x_in, 0, t_in, c_in, coef=coef, bboxs_curr=boxes)


for _ in range(3):
    call()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=shapes) as prof:
    for _ in range(3):
        call()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=shapes).table(sort_by="self_cuda_time_total", row_limit=50, max_name_column_width=50))
