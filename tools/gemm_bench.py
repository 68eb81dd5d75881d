"""The HIP row GEMM (csrc/sta_gemm.hip) against F.linear at the Linear shapes of one CFG UNet call (64 = 2 x 32 images, 512^2):
max error against an fp32 product of the same 16-bit operands, and microseconds per call of both.
usage: python tools/gemm_bench.py [--dtype fp16|bf16] [--only 640x640@65536,...] [--iters 20] [--no-lib]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-spacetime-attn_amd"))
import torch
import torch.nn.functional as F

from sta import fused

# (K, N, rows, bias, how many per UNet call) from tools/gemm_census.py
SHAPES = [(320, 320, 262144, 1, 10), (320, 640, 262144, 0, 5), (640, 640, 65536, 1, 25), (640, 1280, 65536, 0, 5), (640, 5120, 65536, 1, 5),
          (2560, 640, 65536, 1, 5), (1280, 1280, 16384, 1, 25), (1280, 2560, 16384, 0, 5), (1280, 10240, 16384, 1, 5), (5120, 1280, 16384, 1, 5),
          (1280, 1280, 4096, 1, 5), (960, 320, 262144, 1, 1), (1920, 640, 65536, 1, 1)]


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
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-lib", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    tot = {"hip": 0.0, "lib": 0.0}
    for K, N, R, hb, cnt in SHAPES:
        name = "%dx%d@%d" % (K, N, R)
        if a.only and name not in a.only.split(","):
            continue
        x = torch.randn(R // 1024, 1024, K, generator=g).to(dt).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).to(dev)
        b = (0.1 * torch.randn(N, generator=g)).to(dt).to(dev) if hb else None
        with torch.no_grad():
            assert fused.linear_rows_supported(x, w), name
            wp = fused.pack_linear_weight(w)
            ours = lambda: fused.linear_rows(x, wp, N, bias=b)
            lib_ = lambda: F.linear(x, w, b)
            rec = {"shape": name, "per_call": cnt}
            if not a.no_check:
                ref = F.linear(x[:4].float(), w.float(), None if b is None else b.float())
                got = ours()[:4].float()
                rec["max_err"] = round((got - ref).abs().max().item(), 5)
                rec["max_ref"] = round(ref.abs().max().item(), 3)
            flop = 2.0 * R * K * N
            byts = 2.0 * (R * K + R * N + K * N)
            rec["hip_us"] = round(timed(ours, a.iters), 1)
            rec["hip_tflops"] = round(flop / rec["hip_us"] / 1e6, 1)
            rec["hip_gbps"] = round(byts / rec["hip_us"] / 1e3, 1)
            tot["hip"] += cnt * rec["hip_us"]
            if not a.no_lib:
                rec["lib_us"] = round(timed(lib_, a.iters), 1)
                rec["lib_tflops"] = round(flop / rec["lib_us"] / 1e6, 1)
                tot["lib"] += cnt * rec["lib_us"]
        print(json.dumps(rec), flush=True)
    print(json.dumps({"per_unet_call_ms": {k_: round(v / 1e3, 2) for k_, v in tot.items()}}))


if __name__ == "__main__":
    main()
