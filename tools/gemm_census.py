"""Every library GEMM (F.linear / matmul / addmm) of one eager CFG UNet call at the bench batch (32 images = batch 64, fp16, NHWC trunk), timed with an event pair
each: which still go to the library, what they cost including the library's own layout / im2col helper kernels (those run inside the
bracket). usage: python tools/conv_census.py [imgs]"""
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-spacetime-attn_amd"))
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
import torch.nn.functional as F

from sta import fused, prompt_state
from sta.pipeline import build_sd_v1, use_shipped_miopen_db

use_shipped_miopen_db()
imgs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev, dt = torch.device("cuda:0"), torch.float16
model = build_sd_v1(dev, dt, with_vae=False, init_weights=True, seed=0, channels_last=True)
unet = model.model.diffusion_model
from sta.pipeline import DEFAULT_CENTRES, conditionings, load_prompts
r0 = load_prompts(64)[0]
names = (r0["objects"] + ["object"] * 2)[:2]
uc, c, local_c = conditionings(model, r0["prompt"], names, dt)
pair = lambda u, v: torch.stack([u, v], dim=1).reshape(2 * imgs, *u.shape[1:])
c_in = pair(uc.expand(imgs, -1, -1), c.expand(imgs, -1, -1)).contiguous()
x_in = torch.randn(2 * imgs, 4, 64, 64, device=dev)
t_in = torch.full((2 * imgs,), 981, device=dev, dtype=torch.long)
coef = torch.full((imgs, 2), 2.5, device=dev)
boxes = [[list(cc) for cc in DEFAULT_CENTRES[:2]]] * imgs
rec = []
real_conv, real_hip = F.conv2d, fused.conv3x3_nhwc


def timed(kind, fn, key):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    rec.append((kind, key, e0, e1))
    return out


def wrap(name, fn):
    def f(*a, **k):
        ts = [t_ for t_ in a[:3] if torch.is_tensor(t_)]
        key = name + " " + " x ".join("[%s]%s" % (",".join(map(str, t_.shape)), "" if t_.is_contiguous() else "v") for t_ in ts)
        return timed("lib", lambda: fn(*a, **k), key)
    return f


real_linear = F.linear
F.linear = wrap("linear", real_linear)
torch.nn.functional.linear = F.linear
for nm in ("matmul", "bmm", "addmm", "mm", "baddbmm"):
    setattr(torch, nm, wrap(nm, getattr(torch, nm)))
for rep in range(2):
    rec.clear()
    prompt_state.begin_prompt([local_c] * imgs, first_timestep=981)
    with torch.no_grad():
        model.apply_model_extra(x_in, 0, t_in, c_in, coef=coef, bboxs_curr=boxes)
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, key, e0, e1 in rec:
    a = agg.setdefault((kind, key), [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
tot = collections.Counter()
for (kind, key), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-70s %3d calls %9.1f us total %8.1f us each" % (key, n, us, us / n))
    tot[kind] += us
print({k_: round(v / 1e3, 2) for k_, v in tot.items()}, "ms per UNet call")
