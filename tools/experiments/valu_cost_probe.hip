// valu_cost_probe.hip — issue cost of single vector instructions on gfx950 (cycles per wave-instruction at one and two waves per SIMD):
// what a v_exp_f32 costs beside a v_exp_f16, a packed-f32 FMA beside two plain ones, the conversions and the maxima of the softmax of
// csrc/sta_xattn_proj3.hip. Each kernel runs REPS x 64 copies of one instruction over 16 independent registers (no dependences inside a
// group of 16), lane 0 of every wave records s_memtime around the loop. Standalone build (tools/valu_cost_probe.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BODY16(INS) \
  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15)

template <int OP>
__global__ __launch_bounds__(512) void valu_probe(float* out, long long* cyc, int reps) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = -0.001f * (float)(threadIdx.x + 1 + i);
  float r2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r2[i] = 0.5f + 0.01f * i;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int k = 0; k < reps; ++k) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (OP == 0) {
#define I(n) asm volatile("v_exp_f32 %0, %0" : "+v"(r[n]));
        BODY16(I)
#undef I
      } else if constexpr (OP == 1) {
#define I(n) asm volatile("v_exp_f16 %0, %0" : "+v"(r[n]));
        BODY16(I)
#undef I
      } else if constexpr (OP == 2) {
#define I(n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[n]) : "v"(r2[n]));
        BODY16(I)
#undef I
      } else if constexpr (OP == 3) {      // packed f32 FMA on register pairs (8 pairs)
#define I(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&r[2 * (n & 7)]) : "v"(*(double*)&r2[2 * (n & 7)]));
        BODY16(I)
#undef I
      } else if constexpr (OP == 4) {
#define I(n) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[n]) : "v"(r2[n]));
        BODY16(I)
#undef I
      } else if constexpr (OP == 5) {
#define I(n) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r[n]) : "v"(r2[n]));
        BODY16(I)
#undef I
      } else if constexpr (OP == 6) {
#define I(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[n]));
        BODY16(I)
#undef I
      } else if constexpr (OP == 7) {      // v_exp_f16 on the HIGH half through op_sel (VOP3), result to the high half
#define I(n) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r[n]));
        BODY16(I)
#undef I
      } else if constexpr (OP == 8) {
#define I(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&r[2 * (n & 7)]) : "v"(*(double*)&r2[2 * (n & 7)]));
        BODY16(I)
#undef I
      } else if constexpr (OP == 9) {
#define I(n) asm volatile("v_mov_b32 %0, %1" : "+v"(r[n]) : "v"(r2[n]));
        BODY16(I)
#undef I
      } else if constexpr (OP == 10) {
#define I(n) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(r[n]) : "v"(r2[n]));
        BODY16(I)
#undef I
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

extern "C" int valu_cost_probe(int op, int waves_per_wg, float* out, long long* cyc, int reps, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(256), block(64 * waves_per_wg);
  switch (op) {
#define C(n) case n: hipLaunchKernelGGL(valu_probe<n>, grid, block, 0, st, out, cyc, reps); break;
    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10)
#undef C
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
