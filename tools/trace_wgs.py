"""Per-workgroup start/end (100 MHz wall clock) of the forward kernels: how long does the grid take to
start, how long does one workgroup live, when does the last one finish. Debug build (-DSTA_TRACE)."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import lib, ops  # noqa: E402
out = os.path.join(ROOT, "gpurun_out", "libsta_trace.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSTA_TRACE", "-I", lib.INCLUDE, "-I", lib.CSRC, *lib.SOURCES, "-o", out])
lib.LIB_PATH = out
L = lib.load()
L.sta_debug_set_trace.restype, L.sta_debug_set_trace.argtypes = ctypes.c_int, [ctypes.c_void_p]
dev = "cuda"
I = int(sys.argv[1]) if len(sys.argv) > 1 else 1           # images per launch (stamps: image 0's workgroups)
for (N, C) in [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
    K, H, M = 2, 8, 77
    g = torch.Generator().manual_seed(0)
    q = torch.randn(2 * I, N, C, generator=g).bfloat16().to(dev)
    k = torch.randn(I * (K + 2), M, C, generator=g).bfloat16().to(dev)
    v = torch.randn(I * (K + 2), M, C, generator=g).bfloat16().to(dev)
    mask = ops.disc_mask_bits([(0.3, 0.4), (0.7, 0.6)], int(N ** 0.5)).to(dev).repeat(I, 1)
    coef = torch.full((I, K), 2.5, device=dev)
    packed = ops.pack_kv(k, v, H, n_img=I)
    tr = torch.zeros(128 + 2 * 4096, dtype=torch.int64, device=dev)
    tr[0] = 1 << 30      # no per-wave timeline
    tr[1] = 1
    assert L.sta_debug_set_trace(tr.data_ptr()) == 0
    for _ in range(5):
        tr[128:] = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.xattn_forward(q, packed, mask, coef, (C // H) ** -0.5); e1.record()
        torch.cuda.synchronize()
    t = tr[128:].cpu().view(-1, 2)
    t = t[t[:, 0] > 0]
    s, e = t[:, 0], t[:, 1]
    s0 = s.min()
    life = (e - s).float()
    print("N=%d C=%d: %d WGs | first start 0, last start %.2f us | WG life min/med/max %.2f/%.2f/%.2f us | last end %.2f us | event-bracketed %.2f us"
          % (N, C, len(t), (s.max() - s0).item() / 100, life.min() / 100, life.median() / 100, life.max() / 100, (e.max() - s0).item() / 100, e0.elapsed_time(e1) * 1e3))
