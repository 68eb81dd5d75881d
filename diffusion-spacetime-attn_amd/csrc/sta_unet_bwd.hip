// sta_unet_bwd.hip — input gradients of the trunk glue kernels of sta_unet.hip, for the tracked (weight-optimisation) epochs:
// the blend weights are the only leaf, every model parameter is frozen, so each op needs d(input) only — no dgamma / dbeta /
// dW reductions. One or two passes over the activations with 16-byte accesses per lane, fp32 arithmetic: HBM-bound
// (algorithmic bytes: every input read once per pass, dx written once). C-ABI in include/sta_unet.h.
//
// Reference ops differentiated here (PyTorch autograd does it op by op in the reference's tracked epochs):
//   GroupNorm32 -> SiLU                openaimodel.py ResBlock._forward, util.py:216; attention.py:335 (Normalize, no SiLU)
//   GEGLU: x * gelu(gate)              attention.py:43-45
//   LayerNorm of the residual stream   attention.py:274-299

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sta_xattn.h"
#include "sta_unet.h"
#include "sta_internal.h"

namespace {

template <typename T> struct V8T { typedef T type __attribute__((ext_vector_type(8))); };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

int launched(const char* what) {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "%s launch: %s", what, hipGetErrorString(e));
}

// --------------------------------------------------------------------------------------------------
// GroupNorm (+ pre-add) (+ SiLU) backward on NHWC activations, same work split as the forward (sta_unet.hip): a workgroup
// takes a chunk of pixels of one image and all channels, a thread keeps one 8-channel column.
//   xh = (x + add - mean) * rstd;  z = xh * gamma + beta;  y = silu(z) or z
//   t  = dy * silu'(z) * gamma  (silu'(z) = s + z s (1 - s), s = sigmoid z);   per (b, group): m1 = mean t, m2 = mean t xh
//   dx = rstd * (t - m1 - xh * m2)
// Kernel 1 writes per-chunk partial sums of (t, t xh) per group, kernel 2 folds them and writes dx. mean / rstd are
// rebuilt from the FORWARD's partial sums (its workspace, kept by the caller) exactly as the forward folded them.
// --------------------------------------------------------------------------------------------------
constexpr int GN_NT = 512;
constexpr int GN_MAXG = 64;

template <typename T, bool APPLY>
__global__ __launch_bounds__(GN_NT) void gn_nhwc_bwd_kernel(const T* __restrict__ x, const float* __restrict__ add,
                                                           const T* __restrict__ gamma, const T* __restrict__ beta,
                                                           const float* __restrict__ part, const T* __restrict__ dy,
                                                           float* __restrict__ partb, T* __restrict__ dx, int C, int HW, int G,
                                                           int chunk_px, float eps, int silu) {
  using V8 = typename V8T<T>::type;
  __shared__ float mean_s[GN_MAXG], rstd_s[GN_MAXG], m1_s[GN_MAXG], m2_s[GN_MAXG];
  __shared__ float acc[2 * GN_MAXG];
  const int CV = C >> 3, R = GN_NT / CV, Cg = C / G;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int row = threadIdx.x / CV, col = threadIdx.x - row * CV;
  if (threadIdx.x < G) {
    float s = 0.f, q = 0.f, s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < nchunk; ++k) {
      s += part[((size_t)b * nchunk + k) * 2 * G + 2 * threadIdx.x];
      q += part[((size_t)b * nchunk + k) * 2 * G + 2 * threadIdx.x + 1];
      if (APPLY) {
        s1 += partb[((size_t)b * nchunk + k) * 2 * G + 2 * threadIdx.x];
        s2 += partb[((size_t)b * nchunk + k) * 2 * G + 2 * threadIdx.x + 1];
      }
    }
    const float inv_n = 1.0f / ((float)Cg * (float)HW);
    const float mean = s * inv_n;
    mean_s[threadIdx.x] = mean;
    rstd_s[threadIdx.x] = rsqrtf(fmaxf(q * inv_n - mean * mean, 0.f) + eps);
    m1_s[threadIdx.x] = s1 * inv_n;
    m2_s[threadIdx.x] = s2 * inv_n;
  }
  if (!APPLY && threadIdx.x < 2 * G) acc[threadIdx.x] = 0.f;
  __syncthreads();
  if (row < R) {
    float gm[8], bt[8], sh[8], rs[8], m1[8], m2[8], s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col * 8 + e, g = c / Cg;
      const float a = add ? add[(size_t)b * C + c] : 0.f;
      gm[e] = (float)gamma[c];
      bt[e] = (float)beta[c];
      rs[e] = rstd_s[g];
      sh[e] = (a - mean_s[g]) * rs[e];            // xh = x * rstd + sh
      m1[e] = m1_s[g];
      m2[e] = m2_s[g];
      s[e] = 0.f;
      q[e] = 0.f;
    }
    const int p1 = min(HW, (chunk + 1) * chunk_px);
    for (int p = chunk * chunk_px + row; p < p1; p += R) {
      const size_t off = ((size_t)b * HW + p) * C + col * 8;
      const V8 v = *(const V8*)(x + off);
      const V8 d = *(const V8*)(dy + off);
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (float)v[e] * rs[e] + sh[e];
        float t = (float)d[e];
        if (silu) {
          const float z = xh * gm[e] + bt[e];
          const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
          t *= sg * (1.0f + z * (1.0f - sg));
        }
        t *= gm[e];
        if (APPLY) {
          o[e] = (T)(rs[e] * (t - m1[e] - xh * m2[e]));
        } else {
          s[e] += t;
          q[e] += t * xh;
        }
      }
      if (APPLY) *(V8*)(dx + off) = o;
    }
    if (!APPLY) {
      const int g0 = (col * 8) / Cg, g1 = (col * 8 + 7) / Cg;
      float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool first = (col * 8 + e) / Cg == g0;
        s0 += first ? s[e] : 0.f;
        q0 += first ? q[e] : 0.f;
        s1 += first ? 0.f : s[e];
        q1 += first ? 0.f : q[e];
      }
      atomicAdd(&acc[2 * g0], s0);
      atomicAdd(&acc[2 * g0 + 1], q0);
      if (g1 != g0) {
        atomicAdd(&acc[2 * g1], s1);
        atomicAdd(&acc[2 * g1 + 1], q1);
      }
    }
  }
  if (!APPLY) {
    __syncthreads();
    if (threadIdx.x < 2 * G) partb[((size_t)b * nchunk + chunk) * 2 * G + threadIdx.x] = acc[threadIdx.x];
  }
}

// --------------------------------------------------------------------------------------------------
// GEGLU backward: y = a * gelu(g), x = [a | g] per row:  da = dy * gelu(g);  dg = dy * a * (Phi(g) + g phi(g))
// (the same Abramowitz & Stegun 7.1.26 erf as the forward)
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dxo,
                                                      long nvec, int Dv) {
  using V8 = typename V8T<T>::type;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    const long r = i / Dv;
    const int j = (int)(i - r * Dv);
    const V8 a = ((const V8*)x)[r * 2 * Dv + j];
    const V8 gt = ((const V8*)x)[r * 2 * Dv + Dv + j];
    const V8 d = ((const V8*)dy)[i];
    V8 oa, og;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g = (float)gt[e];
      const float ax = fabsf(g) * 0.70710678118654752f;
      const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
      float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
      p = __builtin_fmaf(p, t, 1.421413741f);
      p = __builtin_fmaf(p, t, -0.284496736f);
      p = __builtin_fmaf(p, t, 0.254829592f);
      const float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);     // exp(-g^2 / 2)
      const float er = p * t * ex;                                                  // 1 - erf(|g| / sqrt 2)
      const float phi = g >= 0.f ? 1.0f - 0.5f * er : 0.5f * er;                    // Phi(g)
      const float dv = (float)d[e];
      oa[e] = (T)(dv * g * phi);
      og[e] = (T)(dv * (float)a[e] * (phi + g * 0.3989422804014327f * ex));
    }
    ((V8*)dxo)[r * 2 * Dv + j] = oa;
    ((V8*)dxo)[r * 2 * Dv + Dv + j] = og;
  }
}

// --------------------------------------------------------------------------------------------------
// LayerNorm backward (input gradient), one wave per row, C <= 2048:
//   xh = (s - mean) rstd;  t = dy * gamma;  ds = rstd * (t - mean(t) - xh * mean(t xh)) + dres   (dres: the gradient that
//   reaches the same tensor through the residual connection, may be NULL)
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ sres, const T* __restrict__ gamma,
                                                          const T* __restrict__ dy, const T* __restrict__ dres, T* __restrict__ ds,
                                                          long R, int C, float eps) {
  using V8 = typename V8T<T>::type;
  constexpr int NV = 4;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63, nvec = C >> 3;
  float v[NV][8], t[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = lane + 64 * j;
    if (idx < nvec) {
      const V8 a = ((const V8*)(sres + row * C))[idx];
      const V8 d = ((const V8*)(dy + row * C))[idx];
      const V8 gm = ((const V8*)gamma)[idx];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[j][e] = (float)a[e];
        t[j][e] = (float)d[e] * (float)gm[e];
        sum += v[j][e];
      }
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (lane + 64 * j < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[j][e] -= mean;
        q += v[j][e] * v[j][e];
      }
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (lane + 64 * j < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[j][e] *= rstd;                        // xh
        s1 += t[j][e];
        s2 += t[j][e] * v[j][e];
      }
    }
  const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = lane + 64 * j;
    if (idx < nvec) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rstd * (t[j][e] - m1 - v[j][e] * m2);
      if (dres) {
        const V8 r = ((const V8*)(dres + row * C))[idx];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += (float)r[e];
      }
      V8 ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = (T)o[e];
      ((V8*)(ds + row * C))[idx] = ov;
    }
  }
}

int gn_nhwc_chunks_bwd(int HW) {      // must equal gn_nhwc_chunks of sta_unet.hip (the forward's workspace layout)
  int n = HW / 8;
  if (n > 32) n = 32;
  if (n < 1) n = 1;
  return n;
}

template <typename T>
void gn_bwd_launch(const void* x, const float* add, const void* gamma, const void* beta, const float* part, const void* dy,
                   float* partb, void* dx, int B, int C, int HW, int G, float eps, int silu, hipStream_t st) {
  const int nchunk = gn_nhwc_chunks_bwd(HW), chunk_px = (HW + nchunk - 1) / nchunk;
  const dim3 grid(nchunk, B);
  hipLaunchKernelGGL((gn_nhwc_bwd_kernel<T, false>), grid, dim3(GN_NT), 0, st, (const T*)x, add, (const T*)gamma, (const T*)beta, part,
                     (const T*)dy, partb, (T*)dx, C, HW, G, chunk_px, eps, silu);
  hipLaunchKernelGGL((gn_nhwc_bwd_kernel<T, true>), grid, dim3(GN_NT), 0, st, (const T*)x, add, (const T*)gamma, (const T*)beta, part,
                     (const T*)dy, partb, (T*)dx, C, HW, G, chunk_px, eps, silu);
}

}  // namespace

extern "C" {

int sta_groupnorm_silu_nhwc_bwd(const void* x, const float* add, const void* gamma, const void* beta, const void* dy, void* dx,
                                const void* fwd_workspace, void* bwd_workspace, int B, int C, int HW, int G, float eps, int silu,
                                int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!x || !gamma || !beta || !dy || !dx || !fwd_workspace || !bwd_workspace) return sta_fail(STA_E_ARG, "null pointer");
  if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || G > GN_MAXG || C % G || C % 8 || (C / G < 8 && C / G != 4) || C / 8 > GN_NT)
    return sta_fail(STA_E_ARG, "groupnorm nhwc bwd: B=%d C=%d HW=%d G=%d (need C %% 8 == 0, C/G >= 8 or == 4, C <= %d, G <= %d)", B, C, HW,
                    G, 8 * GN_NT, GN_MAXG);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    gn_bwd_launch<__bf16>(x, add, gamma, beta, (const float*)fwd_workspace, dy, (float*)bwd_workspace, dx, B, C, HW, G, eps, silu, st);
  else
    gn_bwd_launch<_Float16>(x, add, gamma, beta, (const float*)fwd_workspace, dy, (float*)bwd_workspace, dx, B, C, HW, G, eps, silu, st);
  return launched("groupnorm_silu_nhwc_bwd");
}

int sta_geglu_bwd(const void* x, const void* dy, void* dx, long R, int D, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!x || !dy || !dx) return sta_fail(STA_E_ARG, "null pointer");
  if (R <= 0 || D <= 0 || D % 8) return sta_fail(STA_E_ARG, "geglu bwd: R=%ld D=%d (need D %% 8 == 0)", R, D);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const long nvec = R * (D / 8);
  long blocks = (nvec + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(geglu_bwd_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, st, (const __bf16*)x, (const __bf16*)dy, (__bf16*)dx,
                       nvec, D / 8);
  else
    hipLaunchKernelGGL(geglu_bwd_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, st, (const _Float16*)x, (const _Float16*)dy,
                       (_Float16*)dx, nvec, D / 8);
  return launched("geglu_bwd");
}

int sta_layernorm_bwd(const void* s, const void* gamma, const void* dy, const void* dres, void* ds, long R, int C, float eps, int dtype,
                      void* stream) {
  g_sta_err[0] = 0;
  if (!s || !gamma || !dy || !ds) return sta_fail(STA_E_ARG, "null pointer");
  if (R <= 0 || C <= 0 || C % 8 || C > 2048) return sta_fail(STA_E_ARG, "layernorm bwd: R=%ld C=%d (need C %% 8 == 0, C <= 2048)", R, C);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const unsigned blocks = (unsigned)((R + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(layernorm_bwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)s, (const __bf16*)gamma, (const __bf16*)dy,
                       (const __bf16*)dres, (__bf16*)ds, R, C, eps);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, (const _Float16*)s, (const _Float16*)gamma,
                       (const _Float16*)dy, (const _Float16*)dres, (_Float16*)ds, R, C, eps);
  return launched("layernorm_bwd");
}

}  // extern "C"
