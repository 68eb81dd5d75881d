"""How much of a bench step is the VAE decode, per 16-bit type (MIOpen picks different solvers per type)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "diffusion-spacetime-attn_amd")):
    sys.path.insert(0, p)
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
import torch
from sta.pipeline import build_sd_v1
I = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.backends.cudnn.benchmark = os.environ.get("FIND", "1") == "1"      # MIOpen measures its solvers per shape (immediate mode otherwise)
for dt in [getattr(torch, t) for t in os.environ.get("DT", "float16,bfloat16").split(",")]:
    model = build_sd_v1("cuda", dt, with_vae=True, channels_last=True)
    z = torch.randn(I, 4, 64, 64, device="cuda", dtype=dt)
    for cl in (False, True):
        vae = model.first_stage_model
        if cl:
            vae = vae.to(memory_format=torch.channels_last)
        with torch.no_grad():
            for _ in range(2):
                x = model.decode_first_stage(z)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                x = model.decode_first_stage(z)
            torch.cuda.synchronize()
        print(dt, "channels_last" if cl else "nchw", "decode of %d images: %.1f ms" % (I, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    del model
    torch.cuda.empty_cache()
