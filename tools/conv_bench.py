"""The HIP 3x3 convolution (csrc/sta_conv.hip) against the library convolution at the UNet's shapes (64 = 2 x 32 images, NHWC):
max error against an fp32 convolution of the same 16-bit operands, and microseconds per call of both.
usage: python tools/conv_bench.py [--dtype fp16|bf16] [--batch 64] [--only 320x320@64,...] [--iters 20]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-spacetime-attn_amd"))
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
import torch.nn.functional as F

from sta import fused
from sta.pipeline import use_shipped_miopen_db

use_shipped_miopen_db()

# (Cin, Cout, H = W, up2, how many per UNet call) — the 3x3 stride-1 convolutions of SD-v1's UNet at 512^2 the kernel takes
SHAPES = [(320, 320, 64, 0, 7), (640, 320, 64, 0, 2), (960, 320, 64, 0, 1), (640, 640, 64, 1, 1),
          (320, 640, 32, 0, 1), (640, 640, 32, 0, 6), (960, 640, 32, 0, 1), (1280, 640, 32, 0, 1), (1920, 640, 32, 0, 1), (1280, 1280, 32, 1, 1),
          (640, 1280, 16, 0, 1), (1280, 1280, 16, 0, 6), (1920, 1280, 16, 0, 1), (2560, 1280, 16, 0, 2), (1280, 1280, 16, 1, 1),
          (1280, 1280, 8, 0, 11), (2560, 1280, 8, 0, 3)]


def timed(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-lib", action="store_true", help="skip the library convolution (its solver search takes minutes on a fresh box)")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    g = torch.Generator(device="cpu").manual_seed(0)
    tot = {"hip": 0.0, "lib": 0.0, "tflop": 0.0}
    for cin, cout, hw, up2, cnt in SHAPES:
        name = "%dx%d@%d%s" % (cin, cout, hw, "up" if up2 else "")
        if a.only and name not in a.only.split(","):
            continue
        B, Hs = a.batch, hw >> up2
        x = torch.randn(B, cin, Hs, Hs, generator=g).to(dt).to(dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dt).to(dev).contiguous(memory_format=torch.channels_last)
        wp = fused.pack_conv3x3_weight(w)
        with torch.no_grad():
            assert fused.conv3x3_supported(x, w, up2=bool(up2)), name
            ours = lambda: fused.conv3x3_nhwc(x, wp, cout, up2=bool(up2))
            def lib_conv():
                xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if up2 else x
                return F.conv2d(xi, w, None, 1, 1)
            rec = {"shape": name, "per_call": cnt}
            if not a.no_check:
                nb = min(B, 8)
                xi = F.interpolate(x[:nb], scale_factor=2.0, mode="nearest") if up2 else x[:nb]
                ref = F.conv2d(xi.float().contiguous(), w.float().contiguous(), None, 1, 1)
                got = ours()[:nb].float()
                rec["max_err"] = round((got - ref).abs().max().item(), 5)
                rec["max_ref"] = round(ref.abs().max().item(), 3)
            flop = 2.0 * B * hw * hw * 9 * cin * cout
            rec["hip_us"] = round(timed(ours, a.iters), 1)
            rec["hip_tflops"] = round(flop / rec["hip_us"] / 1e6, 1)
            if not a.no_lib:
                rec["lib_us"] = round(timed(lib_conv, a.iters), 1)
                rec["lib_tflops"] = round(flop / rec["lib_us"] / 1e6, 1)
                tot["lib"] += cnt * rec["lib_us"]
            tot["hip"] += cnt * rec["hip_us"]
            tot["tflop"] += cnt * flop / 1e12
        print(json.dumps(rec), flush=True)
    print(json.dumps({"per_unet_call_ms": {k_: round(v / 1e3, 2) for k_, v in tot.items() if k_ != "tflop"}, "tflop": round(tot["tflop"], 2)}))


if __name__ == "__main__":
    main()
