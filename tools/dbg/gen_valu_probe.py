"""Replay a straight-line VALU region of a compiled kernel (literal registers) as a probe: every vector register it touches is loaded
from a random buffer, the region runs (A) as the compiler emitted it and (B) with s_nop 7 behind every instruction, and the registers
it wrote are compared. usage: python tools/dbg/gen_valu_probe.py kernel.s first_line last_line > probe.hip"""
import re, sys
path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lines = [l.strip() for l in open(path).read().split("\n")[a - 1:b]]
ins = [l for l in lines if l and not l.startswith((";", ".")) and not l.endswith(":")]
for l in ins:
    assert not l.startswith(("s_cbranch", "s_branch", "ds_", "buffer_", "global_", "v_mfma", "s_waitcnt", "s_barrier")), l
regs = set()
for l in ins:
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", l):
        regs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", l):
        regs.add(int(m.group(1)))
sregs = set()
for l in ins:
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", l):
        sregs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", l):
        sregs.add(int(m.group(1)))
regs = sorted(regs)
lo, hi = min(regs), max(regs)
base = hi + 8            # address registers above
def body(pad):
    out = []
    # load every register from in + (reg - lo) * 256 + lane * 4
    for r in range(lo, hi + 1):
        out.append("global_load_dword v%d, v%d, %%1 offset:%d" % (r, base, 0))
        out.append("v_add_u32 v%d, 256, v%d" % (base, base))
    out.append("s_waitcnt vmcnt(0)")
    for s in sorted(sregs):
        out.append("s_mov_b32 s%d, 0x3e4ccccd" % s)       # 0.2
    out.append("s_nop 7")
    for l in ins:
        out.append(l)
        if pad:
            out.append("s_nop 7")
    out.append("s_nop 7")
    out.append("s_nop 7")
    for r in range(lo, hi + 1):
        out.append("global_store_dword v%d, v%d, %%2" % (base + 1, r))
        out.append("v_add_u32 v%d, 256, v%d" % (base + 1, base + 1))
    out.append("s_waitcnt vmcnt(0)")
    return "\\n\\t".join(out)
clob = ", ".join('"v%d"' % r for r in range(lo, hi + 12)) + "".join(', "s%d"' % s for s in sorted(sregs)) + ', "memory", "vcc", "scc"'
n = hi - lo + 1
print('''#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define NREG %d
template <int PAD>
__global__ __launch_bounds__(256) void probe(const float* in, float* out, int rounds) {
  const unsigned lane = threadIdx.x & 63;
  for (int it = 0; it < rounds; ++it) {
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* src = in + (size_t)((blockIdx.x * 4 + wv + it * 7) %% 64) * NREG * 64;
    float* dst = out + ((size_t)(blockIdx.x * 4 + wv) * rounds + it) * NREG * 64;
    unsigned off = lane * 4, off2 = lane * 4;
    if (PAD == 0)
      asm volatile("v_mov_b32 v%d, %%0\\n\\tv_mov_b32 v%d, %%3\\n\\t%s" :: "v"(off), "s"(src), "s"(dst), "v"(off2) : %s);
    else
      asm volatile("v_mov_b32 v%d, %%0\\n\\tv_mov_b32 v%d, %%3\\n\\t%s" :: "v"(off), "s"(src), "s"(dst), "v"(off2) : %s);
  }
}
int main() {
  const int WG = 256, rounds = 64;
  size_t nin = (size_t)64 * NREG * 64, nout = (size_t)WG * 4 * rounds * NREG * 64;
  float* h = (float*)malloc(nin * 4);
  srand(5);
  for (size_t i = 0; i < nin; ++i) h[i] = ((rand() %% 20001) - 10000) * 3e-4f;
  float *din, *da, *db;
  hipMalloc(&din, nin * 4); hipMalloc(&da, nout * 4); hipMalloc(&db, nout * 4);
  hipMemcpy(din, h, nin * 4, hipMemcpyHostToDevice);
  float* ha = (float*)malloc(nout * 4); float* hb = (float*)malloc(nout * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe<0>, dim3(WG), dim3(256), 0, 0, din, da, rounds);
    hipLaunchKernelGGL(probe<1>, dim3(WG), dim3(256), 0, 0, din, db, rounds);
    hipDeviceSynchronize();
    hipMemcpy(ha, da, nout * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, nout * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; int perreg[NREG] = {0}; int perrow[4] = {0};
    for (size_t i = 0; i < nout; ++i) if (memcmp(ha + i, hb + i, 4)) { ++bad; perreg[(i / 64) %% NREG]++; perrow[(i %% 64) / 16]++; }
    printf("rep %%d: %%zu of %%zu register-lane values differ between the compiler's spacing and s_nop 7 everywhere [%%s]; by lane row: %%d %%d %%d %%d\\n", rep, bad, nout, hipGetErrorString(hipGetLastError()), perrow[0], perrow[1], perrow[2], perrow[3]);
    for (int r = 0; r < NREG; ++r) if (perreg[r]) printf("   v%%d: %%d\\n", r + %d, perreg[r]);
  }
  return 0;
}''' % (n, base, base + 1, body(False), clob, base, base + 1, body(True), clob, lo))
