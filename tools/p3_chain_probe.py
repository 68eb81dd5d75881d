"""Timing probe of the level-0 head-pair kernel's ATTENTION PHASE in three geometries (tools/experiments/p3_chain_probe.hip: built
from the product's own device code): does a second softmax chain per wave — one wave per SIMD, two 16-pixel groups per wave — run
the phase faster than the product's two waves per SIMD x one group?  VERDICT r04 item 2 / DESIGN section 8 next (2).
    python tools/p3_chain_probe.py [--reps 256] [--dtype fp16]        (on the GPU box; builds build/p3_chain_probe.so if missing)"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "build", "p3_chain_probe.so")
SRC = os.path.join(ROOT, "tools", "experiments", "p3_chain_probe.hip")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffinite-math-only", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "diffusion-spacetime-attn_amd", "csrc"), "-shared", SRC, "-o", SO])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=256, help="items (4 context-heads each) per wave and launch")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--build-only", action="store_true")
    a = ap.parse_args()
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        build()
    if a.build_only:
        return
    L = ctypes.CDLL(SO)
    L.p3_chain_probe.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    kv = (torch.rand(4 * 2 * 2 * 13952 // 2 + 4096, generator=g) * 0.5 + 0.25).to(dt).to(dev)      # positive: finite denominators
    q = (torch.randn(256 * 8 * 2 * 64 * 40, generator=g) * 0.5).to(dt).to(dev)
    sink = torch.zeros(256 * 8 * 64 * 4, dtype=torch.int32, device=dev)
    sl2e = 40 ** -0.5 * 1.4426950408889634
    names = {0: "1 group x 8 waves (product: two waves per SIMD)", 1: "1 group x 4 waves (one wave per SIMD)",
             2: "2 groups x 4 waves (one wave per SIMD, two chains per wave)", 3: "1 group x 8 waves, grouped-MFMA code path",
             4: "2 groups x 8 waves (two waves per SIMD, two chains per wave)",
             5: "1 group x 8 waves, the two mandatory contexts as interleaved chains (6 vector instructions behind every MFMA)",
             6: "1 group x 8 waves, interleaved chains (4 behind every MFMA)",
             7: "1 group x 4 waves (one wave per SIMD), interleaved chains (6 behind every MFMA)",
             8: "1 group x 8 waves, optimistic softmax: no running maximum, scores in log2 units",
             9: "1 group x 8 waves, optimistic softmax + the denominator-class guard"}
    # context-heads per SIMD and launch: waves per SIMD x groups x 4 x reps
    per_simd = {0: 2 * 1 * 4, 1: 1 * 1 * 4, 2: 1 * 2 * 4, 3: 2 * 1 * 4, 4: 2 * 2 * 4, 5: 2 * 1 * 4, 6: 2 * 1 * 4, 7: 1 * 1 * 4, 8: 2 * 1 * 4, 9: 2 * 1 * 4}
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for rnd in range(3):
        for v in sorted(per_simd):
            for _ in range(2):
                assert L.p3_chain_probe(kv.data_ptr(), q.data_ptr(), sink.data_ptr(), a.reps, sl2e, v, int(dt == torch.bfloat16), st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                L.p3_chain_probe(kv.data_ptr(), q.data_ptr(), sink.data_ptr(), a.reps, sl2e, v, int(dt == torch.bfloat16), st)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) * 1e3 / 5)
    out = {"reps": a.reps, "dtype": a.dtype, "variants": {}}
    for v, us in res.items():
        best = min(us)
        out["variants"][names[v]] = {"us": [round(u, 1) for u in us], "ns_per_context_head_per_simd": round(best * 1e3 / (per_simd[v] * a.reps), 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
