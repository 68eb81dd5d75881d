// Issue rate of the MFMA shapes the cross-attention kernels choose between (gfx950): cycles per instruction on one SIMD,
// one wave per SIMD, N_ACC independent accumulators round-robin (dependent-chain latency hidden). s_memtime brackets.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shapes.hip -o gpurun_out/mfma_shapes && gpurun_out/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ void k(float* out, long long* cyc, int iters) {
  f16x8 a8, b8; f16x4 a4, b4;
  for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(0.001f * (threadIdx.x + j)); b8[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
  for (int j = 0; j < 4; ++j) { a4[j] = a8[j]; b4[j] = b8[j]; }
  f32x4 c[8] = {};
  f32x16 d[4] = {};
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (SHAPE == 0) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[u], 0, 0, 0);
      if constexpr (SHAPE == 1) c[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[u], 0, 0, 0);
      if constexpr (SHAPE == 2) d[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d[u & 3], 0, 0, 0);
      if constexpr (SHAPE == 3) d[u & 3] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d[u & 3], 0, 0, 0);
      if constexpr (SHAPE == 4) c[u] = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, c[u], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int u = 0; u < 8; ++u) s += c[u][0];
  for (int u = 0; u < 4; ++u) s += d[u][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[SHAPE] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMallocManaged(&cyc, 8 * sizeof(long long));
  const int iters = 4096;
  const char* names[5] = {"16x16x32_f16", "16x16x16_f16", "32x32x16_f16", "32x32x8_f16", "4x4x4_f16"};
  const double flops[5] = {16384, 8192, 32768, 16384, 512};
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
  }
  for (int s = 0; s < 5; ++s) {
    const double per = (double)cyc[s] / (iters * 8.0);
    printf("%-14s %7.2f cycles / instruction / SIMD   %8.1f flop/cycle/SIMD\n", names[s], per, flops[s] / per);
  }
  return 0;
}
