"""Time the VAE decode of one bench step (32 images, 64x64 latents -> 512x512) with the decoder in NCHW and in NHWC,
next to one CFG UNet call, so that the decode's share of a step can be read off. Usage: python tools/vae_layout_bench.py [imgs] [dtype]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffusion-spacetime-attn_amd"))
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch

from sta.pipeline import build_sd_v1


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)


def main():
    imgs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[2] if len(sys.argv) > 2 else "fp16"]
    dev = torch.device("cuda:0")
    model = build_sd_v1(dev, dt, with_vae=True, init_weights=True, seed=0, channels_last=True)
    vae = model.first_stage_model
    g = torch.Generator(device=dev).manual_seed(1)
    z = torch.randn(imgs, 4, 64, 64, device=dev, dtype=torch.float32, generator=g)
    with torch.no_grad():
        for chunk in (imgs, 8):
            def run():
                return torch.cat([model.decode_first_stage(z[i:i + chunk]) for i in range(0, imgs, chunk)])
            t0 = time.perf_counter()
            ref = run()
            torch.cuda.synchronize()
            print("NCHW decoder, %d images in chunks of %d: first call (solver search) %.1f s" % (imgs, chunk, time.perf_counter() - t0), flush=True)
            print("NCHW decoder, chunks of %d: ms per decode of %d images %s" % (chunk, imgs, ["%.1f" % t for t in timed(run)]), flush=True)
        vae.to(memory_format=torch.channels_last)
        for chunk in (imgs, 8):
            def run():
                return torch.cat([model.decode_first_stage(z[i:i + chunk]) for i in range(0, imgs, chunk)])
            t0 = time.perf_counter()
            out = run()
            torch.cuda.synchronize()
            print("NHWC decoder, chunks of %d: first call %.1f s; max |diff| vs NCHW %.4g (max |ref| %.3g)"
                  % (chunk, time.perf_counter() - t0, (out.float() - ref.float()).abs().max().item(), ref.float().abs().max().item()), flush=True)
            print("NHWC decoder, chunks of %d: ms per decode of %d images %s" % (chunk, imgs, ["%.1f" % t for t in timed(run)]), flush=True)


if __name__ == "__main__":
    main()
