"""In-kernel timeline of the projection-fused forward kernel (debug build with -DSTA_TRACE, never shipped):
per wave of one workgroup, shader-cycle deltas from kernel start to
 1 LDS image landed + barrier | 2 tile 1 begins | 3 tile-1 projection MFMAs issued | 4 ctx0 done | 5 ctx1 done |
 6 local contexts done | 7 stores issued | 8 all tiles done
(head-pair kernel, `pair` as 4th argument: 1 image landed | 2 tile 1 begins | 3 projection issued | 4 head A done | 5 head B done | 8 end)
usage: trace_proj.py [images_per_launch] [wg,wg,...] [waves] [pair]"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import lib, ops  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "libsta_trace.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSTA_TRACE", "-ffinite-math-only",
                       "-I", lib.INCLUDE, "-I", lib.CSRC,
                       *[os.path.join(ROOT, "tools", "experiments", "sta_xattn_proj3_ablate.hip") if s_.endswith("sta_xattn_proj3.hip") else s_ for s_ in lib.SOURCES],
                       "-o", out])      # the head-pair kernel's timeline points live in the experiment build of its source
lib.LIB_PATH = out
L = lib.load()
P3 = len(sys.argv) > 4 and sys.argv[4] == "pair"    # 1 image landed | 2 item 1 begins | 3 projection issued | 4 head A | 5 head B | 8 end | 9-12 after k-steps 1,3,5,7 of item 1
PAIR = False
set_trace = L.sta_debug_set_trace_p3 if P3 else L.sta_debug_set_trace_proj
set_trace.restype, set_trace.argtypes = ctypes.c_int, [ctypes.c_void_p]

dev = "cuda"
I = int(sys.argv[1]) if len(sys.argv) > 1 else 16
WGS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 37, 200]
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N, C, K, H, M = 4096, 320, 2, 8, 77
g = torch.Generator().manual_seed(0)
y = torch.randn(2 * I, N, C, generator=g).half().to(dev)
wq = (torch.randn(C, C, generator=g) / C ** 0.5).half().to(dev)
k = torch.randn(I * (K + 2), M, C, generator=g).half().to(dev)
v = torch.randn(I * (K + 2), M, C, generator=g).half().to(dev)
mask = ops.disc_mask_bits([(0.3, 0.4), (0.7, 0.6)], 64).to(dev).repeat(I, 1)
coef = torch.full((I, K), 2.5, device=dev)
packed, wqf = ops.pack_kv_proj(k, v, H, n_img=I), ops.pack_wq(wq, H)
lib.set_option(lib.OPT_STAGED_WAVES, NW)
lib.set_option(lib.OPT_PROJ_PAIR, 1 if P3 else 2)
for wg in WGS:
    tr = torch.zeros(8 + 16 * 16, dtype=torch.int64, device=dev)
    tr[0] = wg
    assert set_trace(tr.data_ptr()) == 0
    for _ in range(3):
        ops.xattn_forward_proj(y, wqf, packed, mask, coef, (C // H) ** -0.5)
    torch.cuda.synchronize()
    full = tr[8:].cpu().view(16, 16)[:NW]
    wall = (full[:, 14] - full[:, 15]).tolist()
    t = full[:, :14]
    print("%s N=%d C=%d I=%d waves=%d wg=%d" % ("pair" if P3 else "proj", N, C, I, NW, wg))
    base = min(t[w, 0].item() for w in range(NW))
    for w in range(NW):
        row = t[w].tolist()
        mhz = (row[8] - row[0]) / max(wall[w], 1) * 100 if row[8] else 0
        print("  wave %2d start+%5d (%5d ticks@100MHz => %4.0f MHz):" % (w, row[0] - base, wall[w], mhz),
              " ".join("%7d" % (row[i] - row[0]) if row[i] else "      -" for i in range(1, 13 if P3 else 9)))
