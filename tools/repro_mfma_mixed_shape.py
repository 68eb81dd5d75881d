"""The pair hipcc 7.2 does not pad on gfx950, reduced (profiles/r05_hazard_table.md): a 16x16x16 MFMA that accumulates into the result of
a 16x16x32 MFMA fewer than 5 wait states behind it reads registers 0 and 1 of the tile before they are written.

  part 1 (no GPU): a 12-line HIP kernel — k = 32 step, one vector instruction, k = 16 step of the same accumulate chain — compiled with
          hipcc -O3; sta/isa_lint.py shows the distance hipcc left between the two MFMAs (1 wait state, no s_nop).
  part 2 (GPU):    the same three instructions as one asm statement, with 0 .. 7 wait states of s_nop between the MFMAs, against the
          two products computed separately.
usage: python tools/repro_mfma_mixed_shape.py [--gpu]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import isa_lint  # noqa: E402

SRC = r'''
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
extern "C" __global__ void chain(const f16x8* a, const f16x8* b, const f16x4* a2, const f16x4* b2, f32x4* out, unsigned* flag) {
  const int l = threadIdx.x;
  f16x8 av = a[l], bv = b[l];
  f16x4 a2v = a2[l], b2v = b2[l];
  unsigned x = flag[l];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);      // keys 0..31 of a PV chain
  x ^= 0x80000000u;                                                          // one independent vector instruction
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a2v, b2v, acc, 0, 0, 0);       // keys 32..47: accumulates into the k = 32 result
  out[l] = acc;
  flag[l] = x;
}
// the same chain with k wait states of s_nop between the two MFMAs, nothing left to the compiler
template <int K> __device__ void probe(const f16x8 av, const f16x8 bv, const f16x4 a2v, const f16x4 b2v, f32x4* out) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t.rept %5\n\ts_nop 0\n\t.endr\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n\ts_nop 15"
               : "+v"(acc) : "v"(av), "v"(bv), "v"(a2v), "v"(b2v), "n"(K));
  *out = acc;
}
extern "C" __global__ void probes(const f16x8* a, const f16x8* b, const f16x4* a2, const f16x4* b2, f32x4* out) {
  const int l = threadIdx.x;
  probe<0>(a[l], b[l], a2[l], b2[l], out + 0 * 64 + l); probe<1>(a[l], b[l], a2[l], b2[l], out + 1 * 64 + l);
  probe<2>(a[l], b[l], a2[l], b2[l], out + 2 * 64 + l); probe<3>(a[l], b[l], a2[l], b2[l], out + 3 * 64 + l);
  probe<4>(a[l], b[l], a2[l], b2[l], out + 4 * 64 + l); probe<5>(a[l], b[l], a2[l], b2[l], out + 5 * 64 + l);
  probe<6>(a[l], b[l], a2[l], b2[l], out + 6 * 64 + l); probe<16>(a[l], b[l], a2[l], b2[l], out + 7 * 64 + l);
}
'''


def main():
    d = tempfile.mkdtemp(prefix="sta_repro_")
    src = os.path.join(d, "chain.hip")
    open(src, "w").write(SRC)
    asm = isa_lint.compile_to_asm(src, [], [], workdir=d)
    text = open(asm).read()
    body = text[text.index("chain:"):]
    body = body[:body.index("s_endpgm")]
    lines = [l.strip() for l in body.splitlines() if l.strip().startswith(("v_mfma", "v_xor", "s_nop"))]
    print("hipcc -O3 emitted, between the loads and the store of `chain`:")
    for l in lines:
        print("   ", l)
    f = [x for x in isa_lint.lint_text(text, only="^chain$") if "another shape" in x.rule]
    print("sta/isa_lint.py:", isa_lint.format_findings(f) if f else "no finding (this compiler pads the pair)")
    if "--gpu" not in sys.argv:
        return
    import torch
    co = os.path.join(d, "chain.co")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "--cuda-device-only", "--no-gpu-bundle-output", "-o", co, src])
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipModuleLoadData(ctypes.byref(mod), open(co, "rb").read()) == 0
    assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"probes") == 0
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(64, 8, generator=g).half().cuda(), torch.randn(64, 8, generator=g).half().cuda()
    a2, b2 = torch.randn(64, 4, generator=g).half().cuda(), torch.randn(64, 4, generator=g).half().cuda()
    out = torch.zeros(8, 64, 4, device="cuda")
    ptrs = [ctypes.c_void_p(t.data_ptr()) for t in (a, b, a2, b2, out)]
    args = (ctypes.c_void_p * 5)(*[ctypes.cast(ctypes.pointer(p), ctypes.c_void_p) for p in ptrs])
    assert hip.hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, None, args, None) == 0
    torch.cuda.synchronize()
    ref = out[7]                                   # 17 wait states between the two MFMAs
    for k in range(7):
        bad = (out[k] != ref)
        print("k = %d wait states: %3d of 256 values differ from the 17-state chain; registers %s; largest |difference| %.3g"
              % (k, int(bad.sum()), sorted(set(bad.nonzero()[:, 1].tolist())), float((out[k] - ref).abs().max())))


if __name__ == "__main__":
    main()
