"""In-kernel timeline of the fused forward kernel (debug build with -DSTA_TRACE, never shipped).

Builds csrc/sta_xattn.hip with -DSTA_TRACE into gpurun_out/libsta_trace.so, launches the kernel at the
four level shapes and prints, per wave of one workgroup, s_memtime deltas (shader cycles) between:
 (staged kernel: 1 prologue issued | 2 tile bits + local staging issued | 3 DMA landed + barrier | 4 ctx0 done | 5 ctx1 done | 6 all ctx
  | 7 tile-0 stores issued | 8 end of all tiles | 9 tile-1 q requested | 10 ctx0 | 11 ctx1 | 12 all ctx of tile 1)
 usage: trace_fwd.py [images_per_launch] [wg,wg,...]
 0 start | 1 ctx loop entered (mask known for local waves) | 2 K loads issued | 3 S MFMAs issued |
 4 V loads issued | 5 softmax done | 6 PV done, partial in LDS | 7 after barrier | 8 stores issued
"""
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
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSTA_TRACE",
                       "-I", lib.INCLUDE, "-I", lib.CSRC, *lib.SOURCES, "-o", out])
lib.LIB_PATH = out
L = lib.load()
L.sta_debug_set_trace.restype, L.sta_debug_set_trace.argtypes = ctypes.c_int, [ctypes.c_void_p]

dev = "cuda"
I = int(sys.argv[1]) if len(sys.argv) > 1 else 1           # images per launch
WGS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 37]
NW = 8
for (N, C) in [tuple(int(v) for v in t.split('x')) for t in os.environ.get('SHAPES', '4096x320,1024x640').split(',')]:
    K, H, M = 2, 8, 77
    g = torch.Generator().manual_seed(0)
    q = torch.randn(2 * I, N, C, generator=g).bfloat16().to(dev)
    k = torch.randn(I * (K + 2), M, C, generator=g).bfloat16().to(dev)
    v = torch.randn(I * (K + 2), M, C, generator=g).bfloat16().to(dev)
    dim = int(N ** 0.5)
    mask = ops.disc_mask_bits([(0.3, 0.4), (0.7, 0.6)], dim).to(dev).repeat(I, 1)
    coef = torch.full((I, K), 2.5, device=dev)
    packed = ops.pack_kv(k, v, H, n_img=I)
    nwg_guess = 0
    for wg in WGS:
        tr = torch.zeros(8 + NW * 16, dtype=torch.int64, device=dev)
        tr[0] = wg
        assert L.sta_debug_set_trace(tr.data_ptr()) == 0
        for _ in range(3):
            ops.xattn_forward(q, packed, mask, coef, (C // H) ** -0.5)
        torch.cuda.synchronize()
        full = tr[8:].cpu().view(NW, 16)
        wall = (full[:, 14] - full[:, 15]).tolist()     # 100 MHz ticks start -> end
        t = full[:, :14]
        print("N=%d C=%d wg=%d" % (N, C, wg))
        live = [w for w in range(NW) if t[w, 0].item()]
        base = min(t[w, 0].item() for w in live)
        for w in live:
            row = t[w].tolist()
            mhz = (row[8] - row[0]) / max(wall[w], 1) * 100 if row[8] else 0
            print("  wave %d start+%5d (%4d ticks@100MHz => %4.0f MHz):" % (w, row[0] - base, wall[w], mhz), " ".join("%6d" % (row[i] - row[0]) if row[i] else "     -" for i in range(1, 14)))
