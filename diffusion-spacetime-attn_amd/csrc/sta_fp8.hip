// sta_fp8.hip — row-wise dynamic quantisation of 16-bit activations to OCP fp8 (e4m3fn) for the fp8-weight GEMMs of
// BASELINE configs[4] ("fp8 UNet weights on CDNA4 fp8 MFMA"). gfx950 only: v_cvt_pk_fp8_f32 produces the OCP encoding
// there (MI300's fnuz format is a different, incompatible one). C-ABI in include/sta_unet.h.
//
// What it stands in front of (reference ldm/modules/attention.py): the bias-free / biased nn.Linear layers of the
// transformer blocks — to_q/to_k/to_v (:164-166), to_out (:168-171), GEGLU.proj (:50) and FeedForward's output
// Linear (:69) — whose GEMMs then run as e4m3 x e4m3 -> fp32 on the fp8 MFMA path (hipBLASLt through
// torch._scaled_mm) with one fp32 scale per activation row and one per output channel of the weight.
//
// HBM-bound by construction: every element is read once (16 bit) and written once (8 bit); the row stays in
// registers between the amax reduction and the conversion. Roofline = 3 bytes per element + 4 bytes per row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sta_unet.h"
#include "sta_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
constexpr int MAXIT = 10;                    // 64 lanes x 8 elements x 10 = 5120 channels per row at most
constexpr float E4M3_MAX = 448.0f;

template <typename V8>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const V8* __restrict__ x, u32x2* __restrict__ xq,
                                                             float* __restrict__ scale, long rows, int chunks) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);     // one wave per row
  if (row >= rows) return;
  const V8* src = x + row * chunks;
  float v[MAXIT][8];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXIT; ++i) {
    const int ch = lane + 64 * i;
    if (64 * i < chunks) {                   // wave-uniform
      V8 t = {};
      if (ch < chunks) t = src[ch];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[i][j] = (float)t[j];
        amax = fmaxf(amax, fabsf(v[i][j]));
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  const float s = amax > 0.f ? amax / E4M3_MAX : 1.0f;
  const float inv = 1.0f / s;
  if (lane == 0) scale[row] = s;
  u32x2* dst = xq + row * chunks;
#pragma unroll
  for (int i = 0; i < MAXIT; ++i) {
    const int ch = lane + 64 * i;
    if (64 * i < chunks && ch < chunks) {
      unsigned lo = 0, hi = 0;               // 4 fp8 values per dword, element j in byte j & 3
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4] * inv, v[i][5] * inv, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][6] * inv, v[i][7] * inv, hi, true);
      dst[ch] = u32x2{lo, hi};
    }
  }
}

}  // namespace

extern "C" int sta_quant_rows_fp8(const void* x, void* xq, float* scale, long rows, int C, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!x || !xq || !scale) return sta_fail(STA_E_ARG, "null pointer");
  if (rows <= 0 || C <= 0) return sta_fail(STA_E_ARG, "rows=%ld C=%d", rows, C);
  if (C % 8 || C > 64 * 8 * MAXIT) return sta_fail(STA_E_UNSUP, "C=%d unsupported (C %% 8 == 0, C <= %d)", C, 64 * 8 * MAXIT);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(quant_rows_fp8_kernel<bf16x8>, grid, dim3(256), 0, st, (const bf16x8*)x, (u32x2*)xq, scale, rows, C / 8);
  else
    hipLaunchKernelGGL(quant_rows_fp8_kernel<f16x8>, grid, dim3(256), 0, st, (const f16x8*)x, (u32x2*)xq, scale, rows, C / 8);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "quant_rows_fp8 launch: %s", hipGetErrorString(e));
}
