// Does a packed-fp32 VOP3P instruction whose destination pair overlaps a source pair read BOTH halves of that source before it
// writes the low half?  v_pk_fma_f32 v[8:9], v[8:9], v[10:11], v[12:13] op_sel_hi:[0,1,1]  broadcasts src0.lo (v8) to both
// lanes: hi = v8 * v11 + v13 must use the OLD v8. hipcc 7.2 allocates such overlaps (sta_xattn_bwd.hip, round 6: the last use of
// a per-pixel scalar in a run of packed FMAs); profiles/r06_bwd.md has the story. Build: hipcc --offload-arch=gfx950 -O1.
// Waves 0..3 run the probe ITERS times and count lanes whose hi result differs from the same arithmetic into a separate
// destination; waves 4..7 (the second wave of every SIMD) run `mode`: 0 idle, 1 MFMA stream, 2 packed-fp32 + transcendental stream,
// 3 the same probe.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float* in, unsigned* bad, int iters) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float a = in[lane], b0 = in[64 + lane], b1 = in[128 + lane], c0 = in[192 + lane], c1 = in[256 + lane];
  unsigned nbad = 0;
  if (wv < 4 || MODE == 3) {
    for (int i = 0; i < iters; ++i) {
      float lo, hi, rlo, rhi;
      asm volatile(
          "v_mov_b32 v8, %4\n\tv_mov_b32 v9, 0x7fc00000\n\t"
          "v_mov_b32 v10, %5\n\tv_mov_b32 v11, %6\n\tv_mov_b32 v12, %7\n\tv_mov_b32 v13, %8\n\t"
          "s_nop 4\n\t"
          "v_pk_fma_f32 v[14:15], v[8:9], v[10:11], v[12:13] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
          "v_pk_fma_f32 v[8:9], v[8:9], v[10:11], v[12:13] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
          "s_nop 4\n\t"
          "v_mov_b32 %0, v8\n\tv_mov_b32 %1, v9\n\tv_mov_b32 %2, v14\n\tv_mov_b32 %3, v15\n\t"
          : "=v"(lo), "=v"(hi), "=v"(rlo), "=v"(rhi)
          : "v"(a), "v"(b0), "v"(b1), "v"(c0), "v"(c1)
          : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
      nbad += (__float_as_uint(hi) != __float_as_uint(rhi)) + (__float_as_uint(lo) != __float_as_uint(rlo));
      a += 1.0f;
    }
  } else if (MODE == 1) {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    typedef __attribute__((ext_vector_type(4))) float f4;
    h8 x = {}, y = {};
    f4 acc[4] = {};
    for (int i = 0; i < iters * 2; ++i)
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[j], 0, 0, 0);
    nbad = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] != 0.f;
  } else if (MODE == 2) {
    float s = a, t = b0;
    for (int i = 0; i < iters * 4; ++i) {
      asm volatile("v_pk_mul_f32 v[20:21], v[20:21], v[22:23]\n\tv_exp_f32 %0, %0\n\tv_pk_add_f32 v[24:25], v[20:21], v[22:23]\n\tv_fma_f32 %1, %1, %0, %1"
                   : "+v"(s), "+v"(t) :: "v20", "v21", "v22", "v23", "v24", "v25");
    }
    nbad = (s + t == 12345.f);
  }
  if (lane == 0 || nbad) atomicAdd(bad + wv, nbad);
}

int main() {
  float h[320];
  for (int i = 0; i < 320; ++i) h[i] = 0.37f * (i % 17) - 2.1f + 0.001f * i;
  float* d; unsigned* bad;
  hipMalloc(&d, sizeof h); hipMalloc(&bad, 8 * 4);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(bad, 0, 32);
    const int iters = 200000;
    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, d, bad, iters);
    if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, d, bad, iters);
    if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, d, bad, iters);
    if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, d, bad, iters);
    unsigned hb[8];
    hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
    printf("aggressor mode %d: mismatching results per wave (of %d x 256 workgroups x 64 lanes x 2):", mode, iters);
    for (int w = 0; w < 8; ++w) printf(" %u", hb[w]);
    printf("  [%s]\n", hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
