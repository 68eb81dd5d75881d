// Do VALU instructions issue under the cover of MFMAs on one SIMD (gfx950)? A wave runs a loop of 8 independent
// v_mfma_f32_16x16x32_f16 (16 passes of the matrix pipe each) with K independent vector instructions after every MFMA, all in
// volatile inline asm so that the order is the program order. Reported: shader cycles per MFMA for K = 0..8, for plain VALU
// (v_fma_f32) and for the transcendental unit (v_exp_f32), at 1, 2 and 3 waves per SIMD (the other waves run the same loop).
//   co-issue works  <=>  cycles per MFMA stay ~16 while K * cost(VALU) <= 16
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_coissue.hip -o build/mfma_valu_coissue && build/mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int K, int KIND, int MFMA>      // MFMA: 0 none, 1 v_mfma_f32_16x16x32_f16, 2 v_mfma_f32_32x32x16_f16 (K vector instructions after each)
__global__ void k(float* out, long long* cyc, int iters, int slot) {
  f16x8 a8, b8;
  for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(0.001f * (threadIdx.x + j)); b8[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
  f32x4 c[8] = {};
  f32x16 d[4] = {};
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.5f + 0.01f * j + 1e-3f * threadIdx.x;
  const float y = 0.999f;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MFMA == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[u]) : "v"(a8), "v"(b8));
      if (MFMA == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d[u & 3]) : "v"(a8), "v"(b8));
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(y));
        if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
        if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[j]) : "v"(y));
        if (KIND == 3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(y));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int u = 0; u < 8; ++u) s += c[u][0];
  for (int u = 0; u < 4; ++u) s += d[u][0];
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }   // every wave of block 0
}

// Split roles: waves 0-3 of the workgroup (one per SIMD) issue only MFMAs, waves 4-7 (the second wave of each SIMD) only K vector
// instructions per MFMA of their partner: does the vector work of ANOTHER wave run under the MFMA cover?
template <int K, int KIND, int SHAPE>
__global__ void ksplit(float* out, long long* cyc, int iters, int slot) {
  f16x8 a8, b8;
  for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(0.001f * (threadIdx.x + j)); b8[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
  f32x4 c[8] = {};
  f32x16 d[4] = {};
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.5f + 0.01f * j + 1e-3f * threadIdx.x;
  const float y = 0.999f;
  const bool mfma_wave = (threadIdx.x >> 6) < 4;
  const long long t0 = __builtin_readcyclecounter();
  if (mfma_wave) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (SHAPE == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[u]) : "v"(a8), "v"(b8));
        if (SHAPE == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d[u & 3]) : "v"(a8), "v"(b8));
      }
    }
  } else {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(y));
          if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
        }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int u = 0; u < 8; ++u) s += c[u][0];
  for (int u = 0; u < 4; ++u) s += d[u][0];
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int K, int KIND, int SHAPE>
void run_split(float* out, long long* cyc, const char* name) {
  const int iters = 2048;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((ksplit<K, KIND, SHAPE>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 0);
    hipDeviceSynchronize();
  }
  double m = 0, v = 0;
  for (int w = 0; w < 4; ++w) m += (double)(cyc[2 * w + 1] - cyc[2 * w]) / 4;
  for (int w = 4; w < 8; ++w) v += (double)(cyc[2 * w + 1] - cyc[2 * w]) / 4;
  printf("  %-26s K=%d: MFMA waves %6.1f cycles per MFMA, vector waves %6.1f cycles per K instructions\n", name, K, m / (iters * 8.0), v / (iters * 8.0));
}

template <int K, int KIND, int MFMA>
double run(float* out, long long* cyc, int waves_per_simd) {
  const int iters = 2048;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<K, KIND, MFMA>), dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters, 0);
    hipDeviceSynchronize();
  }
  long long lo = cyc[0], hi = cyc[1];                 // the block's span: first start to last end (the oldest wave wins the arbitration)
  for (int w = 1; w < 4 * waves_per_simd; ++w) { lo = cyc[2 * w] < lo ? cyc[2 * w] : lo; hi = cyc[2 * w + 1] > hi ? cyc[2 * w + 1] : hi; }
  return (double)(hi - lo) / (iters * 8.0);
}

template <int KIND, int M>
void table(const char* name, float* out, long long* cyc) {
  printf("%s x K after every %s: SIMD cycles per MFMA\n", name, M == 1 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_32x32x16_f16");
  printf("  K :      0      1      2      3      4      6      8 |  K=4 without the MFMAs\n");
  for (int w = 1; w <= 3; ++w) {
    printf("  %d wave%s: %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f | %6.1f\n", w, w > 1 ? "s" : " ", run<0, KIND, M>(out, cyc, w) / w,
           run<1, KIND, M>(out, cyc, w) / w, run<2, KIND, M>(out, cyc, w) / w, run<3, KIND, M>(out, cyc, w) / w,
           run<4, KIND, M>(out, cyc, w) / w, run<6, KIND, M>(out, cyc, w) / w, run<8, KIND, M>(out, cyc, w) / w,
           run<4, KIND, 0>(out, cyc, w) / w);
  }
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMallocManaged(&cyc, 64 * sizeof(long long));
  printf("SIMD cycles per MFMA = (last end - first start over the waves of workgroup 0) / MFMAs per wave / waves per SIMD; every wave runs the same loop\n");
  table<0, 1>("v_fma_f32", out, cyc);
  table<1, 1>("v_exp_f32", out, cyc);
  table<2, 1>("v_cvt_pk_f16_f32", out, cyc);
  table<0, 2>("v_fma_f32", out, cyc);
  table<1, 2>("v_exp_f32", out, cyc);
  {   // effective shader clock under each MFMA shape (whole chip busy, 3 waves per SIMD, MFMA only): cycles of workgroup 0 / wall time
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape = 1; shape <= 2; ++shape) {
      const int iters = 1 << 17;
      float ms = 0.f;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (shape == 1) hipLaunchKernelGGL((k<0, 0, 1>), dim3(256), dim3(768), 0, 0, out, cyc, iters, 0);
        else hipLaunchKernelGGL((k<0, 0, 2>), dim3(256), dim3(768), 0, 0, out, cyc, iters, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      long long lo = cyc[0], hi = cyc[1];
      for (int w = 1; w < 12; ++w) { lo = cyc[2 * w] < lo ? cyc[2 * w] : lo; hi = cyc[2 * w + 1] > hi ? cyc[2 * w + 1] : hi; }
      printf("MFMA-only, 3 waves per SIMD, %s: %.2f ms wall, %lld cycles => %.0f MHz; %.0f TFLOP/s\n", shape == 1 ? "16x16x32" : "32x32x16", ms,
             hi - lo, (hi - lo) / (ms * 1e3), 256.0 * 12 * iters * 8 * (shape == 1 ? 16384.0 : 32768.0) / (ms * 1e-3) / 1e12);
    }
  }
  printf("split roles (waves 0-3 MFMA only, waves 4-7 of the same SIMDs vector only):\n");
  run_split<2, 0, 1>(out, cyc, "16x16x32 | v_fma_f32");
  run_split<4, 0, 1>(out, cyc, "16x16x32 | v_fma_f32");
  run_split<8, 0, 1>(out, cyc, "16x16x32 | v_fma_f32");
  run_split<2, 1, 1>(out, cyc, "16x16x32 | v_exp_f32");
  run_split<4, 0, 2>(out, cyc, "32x32x16 | v_fma_f32");
  run_split<8, 0, 2>(out, cyc, "32x32x16 | v_fma_f32");
  run_split<4, 1, 2>(out, cyc, "32x32x16 | v_exp_f32");
  return 0;
}
