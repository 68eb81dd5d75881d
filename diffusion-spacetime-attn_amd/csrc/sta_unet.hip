// sta_unet.hip — HBM-bound glue kernels of the UNet trunk around the cross-attention path (gfx950).
// C-ABI in include/sta_unet.h. Every kernel is one pass over its activations with 16-byte accesses per lane,
// fp32 arithmetic, and no LDS beyond a few floats for workgroup reductions: the roofline that bounds them is
// HBM bandwidth (algorithmic bytes: read every input once, write every output once).

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

// sum over the workgroup; `sh` holds NT/64 floats; every thread gets the result
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  if constexpr (NT == 64) return v;
  __syncthreads();                       // sh may still be read from a previous reduction
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) t += sh[w];
  return t;
}

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// --------------------------------------------------------------------------------------------------
// GroupNorm (+ per-(b,c) pre-add) (+ SiLU), NCHW. One workgroup per (b, group) slab of Cg*HW contiguous
// elements. CACHED: the slab (<= NT*NV*8 elements) stays in registers between statistics and output.
// --------------------------------------------------------------------------------------------------
template <typename T, int NT, int NV, bool CACHED>
__global__ __launch_bounds__(NT) void gn_silu_kernel(const T* __restrict__ x, const float* __restrict__ add,
                                                    const T* __restrict__ gamma, const T* __restrict__ beta,
                                                    T* __restrict__ y, int C, int HW, int G, float eps, int silu) {
  using V8 = typename V8T<T>::type;
  __shared__ float sh[NT / 64];
  const int Cg = C / G;
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const size_t base = ((size_t)b * C + (size_t)g * Cg) * HW;
  const int n = Cg * HW, nvec = n >> 3;
  const int c0 = g * Cg;
  const V8* xv = (const V8*)(x + base);
  V8* yv = (V8*)(y + base);
  const float* addb = add ? add + (size_t)b * C + c0 : nullptr;
  const float inv_n = 1.0f / (float)n;

  if constexpr (CACHED) {
    float v[NV][8];
    int ch[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = threadIdx.x + j * NT;
      ch[j] = 0;
      if (idx < nvec) {
        const V8 r = xv[idx];
        ch[j] = (idx << 3) / HW;
        const float a = addb ? addb[ch[j]] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[j][e] = (float)r[e] + a;
          s += v[j][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
      }
    }
    const float mean = block_sum<NT>(s, sh) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (threadIdx.x + j * NT < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dlt = v[j][e] - mean;
          q += dlt * dlt;
        }
      }
    const float rstd = rsqrtf(block_sum<NT>(q, sh) * inv_n + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = threadIdx.x + j * NT;
      if (idx < nvec) {
        const float gm = (float)gamma[c0 + ch[j]] * rstd;
        const float bt = (float)beta[c0 + ch[j]] - mean * gm;
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = v[j][e] * gm + bt;
          if (silu) t = silu_f(t);
          o[e] = (T)t;
        }
        yv[idx] = o;
      }
    }
  } else {
    float s = 0.f, q = 0.f;
    for (int idx = threadIdx.x; idx < nvec; idx += NT) {
      const V8 r = xv[idx];
      const float a = addb ? addb[(idx << 3) / HW] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = (float)r[e] + a;
        s += t;
        q += t * t;
      }
    }
    const float mean = block_sum<NT>(s, sh) * inv_n;
    const float var = fmaxf(block_sum<NT>(q, sh) * inv_n - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    for (int idx = threadIdx.x; idx < nvec; idx += NT) {
      const V8 r = xv[idx];                       // second read: L2 / Infinity Cache
      const int ch = (idx << 3) / HW;
      const float a = addb ? addb[ch] : 0.f;
      const float gm = (float)gamma[c0 + ch] * rstd;
      const float bt = (float)beta[c0 + ch] - mean * gm + a * gm;
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = (float)r[e] * gm + bt;
        if (silu) t = silu_f(t);
        o[e] = (T)t;
      }
      yv[idx] = o;
    }
  }
}

// --------------------------------------------------------------------------------------------------
// GroupNorm (+ pre-add) (+ SiLU) on NHWC ("channels_last") activations: x, y [B][HW][C].
// A (b, group) slab is strided here (Cg channels of every pixel), so the work is split the other way: a workgroup
// takes a chunk of pixels of one image and ALL channels — fully coalesced rows — and a thread keeps one fixed
// 8-channel column (16 bytes) of every R-th pixel. Kernel 1 writes per-chunk partial sums per group, kernel 2
// folds them (tiny) and normalises. x is read twice (the second time mostly from L2 / Infinity Cache).
// --------------------------------------------------------------------------------------------------
constexpr int GN_NT = 512;          // threads per workgroup: C / 8 <= 512 columns
constexpr int GN_MAXG = 64;

template <typename T>
__global__ __launch_bounds__(GN_NT) void gn_nhwc_stats_kernel(const T* __restrict__ x, const T* __restrict__ xb, int Ca, const float* __restrict__ add,
                                                             float* __restrict__ part, int C, int HW, int G, int chunk_px) {
  using V8 = typename V8T<T>::type;
  __shared__ float acc[2 * GN_MAXG];
  const int CV = C >> 3, R = GN_NT / CV, Cg = C / G;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int row = threadIdx.x / CV, col = threadIdx.x - row * CV;
  if (threadIdx.x < 2 * G) acc[threadIdx.x] = 0.f;
  __syncthreads();
  if (row < R) {
    float a[8], s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] = add ? add[(size_t)b * C + col * 8 + e] : 0.f;
      s[e] = 0.f;
      q[e] = 0.f;
    }
    const int p1 = min(HW, (chunk + 1) * chunk_px);
    // (xb: the input is the channel concatenation [x | xb] of two tensors with Ca and C - Ca channels, read in place)
    const bool second = xb && col * 8 >= Ca;
    const T* src = second ? xb + (col * 8 - Ca) : x + col * 8;
    const int ld = xb ? (second ? C - Ca : Ca) : C;
    for (int p = chunk * chunk_px + row; p < p1; p += R) {
      const V8 v = *(const V8*)(src + ((size_t)b * HW + p) * ld);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = (float)v[e] + a[e];
        s[e] += t;
        q[e] += t * t;
      }
    }
    // fold the 8 channels into their (at most two: Cg >= 8 or Cg == 4) groups, then one LDS atomic per group and moment
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
  __syncthreads();
  if (threadIdx.x < 2 * G) part[((size_t)b * gridDim.x + chunk) * 2 * G + threadIdx.x] = acc[threadIdx.x];
}

template <typename T>
__global__ __launch_bounds__(GN_NT) void gn_nhwc_apply_kernel(const T* __restrict__ x, const T* __restrict__ xb, int Ca, const float* __restrict__ add,
                                                             const float* __restrict__ cs_a, const float* __restrict__ cs_b,
                                                             const T* __restrict__ gamma, const T* __restrict__ beta,
                                                             const float* __restrict__ part, T* __restrict__ y, int C, int HW,
                                                             int G, int chunk_px, float eps, int silu) {
  using V8 = typename V8T<T>::type;
  __shared__ float mean_s[GN_MAXG], rstd_s[GN_MAXG];
  const int CV = C >> 3, R = GN_NT / CV, Cg = C / G;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int row = threadIdx.x / CV, col = threadIdx.x - row * CV;
  if (threadIdx.x < G) {
    float s = 0.f, q = 0.f;
    if (cs_a) {
      // statistics accumulated per CHANNEL by the producing kernel (sum S_c, sum of squares Q_c of the stored values; csrc/sta_conv.hip,
      // sta_gemm.hip): with the per-(b, c) pre-add a, sum (x + a) = S + HW a and sum (x + a)^2 = Q + 2 a S + HW a^2
      const int Cb = C - Ca;
      for (int c = threadIdx.x * Cg; c < (threadIdx.x + 1) * Cg; ++c) {
        const float* st = (cs_b && c >= Ca) ? cs_b + ((size_t)b * Cb + (c - Ca)) * 2 : cs_a + ((size_t)b * (cs_b ? Ca : C) + c) * 2;
        const float a = add ? add[(size_t)b * C + c] : 0.f;
        s += st[0] + (float)HW * a;
        q += st[1] + 2.f * a * st[0] + (float)HW * a * a;
      }
    } else
    for (int k = 0; k < nchunk; ++k) {
      s += part[((size_t)b * nchunk + k) * 2 * G + 2 * threadIdx.x];
      q += part[((size_t)b * nchunk + k) * 2 * G + 2 * threadIdx.x + 1];
    }
    const float inv_n = 1.0f / ((float)Cg * (float)HW);
    const float mean = s * inv_n;
    mean_s[threadIdx.x] = mean;
    rstd_s[threadIdx.x] = rsqrtf(fmaxf(q * inv_n - mean * mean, 0.f) + eps);
  }
  __syncthreads();
  if (row >= R) return;
  float gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = col * 8 + e, g = c / Cg;
    const float a = add ? add[(size_t)b * C + c] : 0.f;
    gm[e] = (float)gamma[c] * rstd_s[g];
    bt[e] = (float)beta[c] + (a - mean_s[g]) * gm[e];
  }
  const int p1 = min(HW, (chunk + 1) * chunk_px);
  const bool second = xb && col * 8 >= Ca;
  const T* src = second ? xb + (col * 8 - Ca) : x + col * 8;
  const int ld = xb ? (second ? C - Ca : Ca) : C;
  for (int p = chunk * chunk_px + row; p < p1; p += R) {
    const size_t off = ((size_t)b * HW + p) * C + col * 8;
    const V8 v = *(const V8*)(src + ((size_t)b * HW + p) * ld);
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = (float)v[e] * gm[e] + bt[e];
      if (silu) t = silu_f(t);
      o[e] = (T)t;
    }
    *(V8*)(y + off) = o;
  }
}

// partial[b][slot][c][2] (the epilogues of csrc/sta_conv.hip / sta_gemm.hip) -> stats[b][c][2]: fixed summation order, no atomics.
// A workgroup folds 64 consecutive floats of one image: 16 lanes x float4 across, 16 slot groups down (slots k = sg, sg + 16, ...),
// then a 16-way tree through LDS — 16-byte loads, at most slots / 16 of them in a thread's chain.
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats, int slots, int C) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  __shared__ f4 red[16][16];
  const int col = threadIdx.x & 15, sg = threadIdx.x >> 4, b = blockIdx.y;
  const int i = (blockIdx.x * 16 + col) * 4;                 // first of this lane's four floats inside the 2 C of an image
  f4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < 2 * C) {
    const float* src = partial + (size_t)b * slots * 2 * C + i;
    for (int k = sg; k < slots; k += 16) s += *(const f4*)(src + (size_t)k * 2 * C);
  }
  red[sg][col] = s;
  __syncthreads();
  if (sg == 0 && i < 2 * C) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][col];
    *(f4*)(stats + (size_t)b * 2 * C + i) = s;
  }
}

// y = a + b + bias[c] over [rows][C] (NHWC activations or token tensors)
template <typename T>
__global__ __launch_bounds__(256) void add_bias_rows_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                          const T* __restrict__ bias, T* __restrict__ y, long nvec, int CV) {
  using V8 = typename V8T<T>::type;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    const V8 av = ((const V8*)a)[i];
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (float)av[e];
    if (b) {
      const V8 bv = ((const V8*)b)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] += (float)bv[e];
    }
    if (bias) {
      const V8 cv = ((const V8*)bias)[(int)(i % CV)];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] += (float)cv[e];
    }
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (T)t[e];
    ((V8*)y)[i] = o;
  }
}

// --------------------------------------------------------------------------------------------------
// GEGLU: y[r][j] = x[r][j] * gelu(x[r][D + j])
// --------------------------------------------------------------------------------------------------
// exact-erf GELU with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute: 2000x below one ulp of the 16-bit
// result around 1): a reciprocal, five fma and one exp2 instead of libm's branchy erff: 434 -> 410 us at [262144 x 2560] (4.9 TB/s
// of reads + writes), 216 -> 195 us at [65536 x 5120].
__device__ __forceinline__ float gelu_erf(float g) {
  const float x = fabsf(g) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);      // 1 - erf(|g| / sqrt 2)
  const float phi = g >= 0.f ? 1.0f - 0.5f * e : 0.5f * e;                            // Phi(g)
  return g * phi;
}

template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const T* __restrict__ x, T* __restrict__ y, long nvec, int Dv) {
  using V8 = typename V8T<T>::type;
  const long stride = (long)gridDim.x * 256;      // (4 vectors per thread and iteration measured no better: 418 vs 410 us)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    const long r = i / Dv;
    const int j = (int)(i - r * Dv);
    const V8 a = ((const V8*)x)[r * 2 * Dv + j];
    const V8 gt = ((const V8*)x)[r * 2 * Dv + Dv + j];
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (T)((float)a[e] * gelu_erf((float)gt[e]));
    ((V8*)y)[i] = o;
  }
}

// --------------------------------------------------------------------------------------------------
// s = x + f + bias; y = LayerNorm(s) * gamma + beta. One wave per row, C <= 2048.
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const T* __restrict__ x, const T* __restrict__ f,
                                                          const T* __restrict__ bias, const T* __restrict__ gamma,
                                                          const T* __restrict__ beta, T* s_out, T* __restrict__ y,
                                                          long R, int C, float eps) {
  using V8 = typename V8T<T>::type;
  constexpr int NV = 4;                                   // vectors per lane: C <= 64 * 4 * 8
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63, nvec = C >> 3;
  const V8* xr = (const V8*)(x + row * C);
  const V8* fr = f ? (const V8*)(f + row * C) : nullptr;
  float v[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = lane + 64 * j;
    if (idx < nvec) {
      const V8 a = xr[idx];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = (float)a[e];
      if (fr) {
        const V8 b = fr[idx];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] += (float)b[e];
      }
      if (bias) {
        const V8 c = ((const V8*)bias)[idx];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] += (float)c[e];
      }
      if (s_out) {   // the residual stream continues in the activation dtype: normalise what is stored
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = (T)v[j][e];
          v[j][e] = (float)o[e];
        }
        ((V8*)(s_out + row * C))[idx] = o;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[j][e];
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (lane + 64 * j < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dlt = v[j][e] - mean;
        q += dlt * dlt;
      }
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = lane + 64 * j;
    if (idx < nvec) {
      const V8 gm = ((const V8*)gamma)[idx], bt = ((const V8*)beta)[idx];
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (T)((v[j][e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
      ((V8*)(y + row * C))[idx] = o;
    }
  }
}

// --------------------------------------------------------------------------------------------------
// The same pass with y written in QUERY-FRAGMENT order — the layout sta_xattn_fwd_proj_qfrag consumes. y = norm2(x) at
// SD-v1 level 0 has exactly one consumer, the projection-fused cross-attention kernel, whose MFMA B operand wants lane
// (g, c) of a wave to hold 16 bytes of pixel c: read from row-major [R][C] that is 16 half-used 128-byte lines per load
// instruction (the shape profiles/r03_vmem_shapes.txt prices at 1.5x the adjacent-lane one). Here the 16 rows of a pixel
// group leave as C/32 fragments of 1 KiB: fragment s holds, at byte (16 g + c) * 16, channels 32 s + 8 g .. + 7 of row c —
// a transpose of [16 rows][C/8 chunks] to [C/8 chunks][16 rows] in 16-byte units, (R/16) * (C/32) KiB in all (the same
// bytes as row-major). One workgroup = one group of 16 rows, a wave takes 4 of them (same per-row arithmetic and lane
// assignment as add_layernorm_kernel: bit-identical values); the chunks cross LDS (272-byte chunk stride: conflict-free
// 16-byte writes and reads) so that the stores are 1-KiB contiguous per wave instruction.
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_qfrag_kernel(const T* __restrict__ x, const T* __restrict__ f,
                                                                const T* __restrict__ bias, const T* __restrict__ gamma,
                                                                const T* __restrict__ beta, T* s_out, T* __restrict__ y,
                                                                long R, int C, float eps) {
  using V8 = typename V8T<T>::type;
  constexpr int CHUNK = 272;                              // LDS bytes per chunk: 16 rows x 16 B + 16 B of padding
  __shared__ __attribute__((aligned(16))) char tile[64 * CHUNK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nvec = C >> 3;
  const long row0 = (long)blockIdx.x * 16 + 4 * wv;
  const bool on = lane < nvec;
  float v[4][8];
  float sum[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {                           // every load of the wave's four rows is issued before the first use
    const long row = row0 + r;
    V8 a = {}, b = {};
    if (on) {
      a = ((const V8*)(x + row * C))[lane];
      if (f) b = ((const V8*)(f + row * C))[lane];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[r][e] = (float)a[e];
    if (f) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[r][e] += (float)b[e];
    }
  }
  V8 bs = {}, gm = {}, bt = {};
  if (on) {
    if (bias) bs = ((const V8*)bias)[lane];
    gm = ((const V8*)gamma)[lane];
    bt = ((const V8*)beta)[lane];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long row = row0 + r;
    if (bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[r][e] += (float)bs[e];
    }
    if (s_out) {
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (T)v[r][e];
        v[r][e] = (float)o[e];
      }
      if (on) ((V8*)(s_out + row * C))[lane] = o;
    }
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) t += v[r][e];
    sum[r] = on ? t : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float mean = wave_sum(sum[r]) / (float)C;
    float q = 0.f;
    if (on) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dlt = v[r][e] - mean;
        q += dlt * dlt;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (T)((v[r][e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
    if (on) *(V8*)(tile + lane * CHUNK + (4 * wv + r) * 16) = o;
  }
  __syncthreads();
  V8* yo = (V8*)y + (size_t)blockIdx.x * nvec * 16;
  for (int t = threadIdx.x; t < nvec * 16; t += 256) yo[t] = *(const V8*)(tile + (t >> 4) * CHUNK + (t & 15) * 16);
}

// The same for 512 < C <= 1024 (SD-v1 level 1, C = 640: norm2's output feeding the locals-from-L2 projection-fused kernel of
// csrc/sta_xattn_proj.hip): a lane holds chunks `lane` and `lane + 64` of its four rows — add_layernorm_kernel's own lane
// assignment and summation order (chunk 0's eight values, then chunk 1's), so the values stay bit-identical to the row-major pass.
template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_qfrag_wide_kernel(const T* __restrict__ x, const T* __restrict__ f,
                                                                     const T* __restrict__ bias, const T* __restrict__ gamma,
                                                                     const T* __restrict__ beta, T* s_out, T* __restrict__ y,
                                                                     long R, int C, float eps) {
  using V8 = typename V8T<T>::type;
  constexpr int CHUNK = 272, NV = 2;
  __shared__ __attribute__((aligned(16))) char tile[64 * NV * CHUNK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nvec = C >> 3;
  const long row0 = (long)blockIdx.x * 16 + 4 * wv;
  float v[4][NV][8];
  float sum[4], sq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long row = row0 + r;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = lane + 64 * j;
      V8 a = {}, b = {};
      if (idx < nvec) {
        a = ((const V8*)(x + row * C))[idx];
        if (f) b = ((const V8*)(f + row * C))[idx];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[r][j][e] = (float)a[e];
      if (f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[r][j][e] += (float)b[e];
      }
    }
  }
  V8 bs[NV] = {}, gm[NV] = {}, bt[NV] = {};
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = lane + 64 * j;
    if (idx < nvec) {
      if (bias) bs[j] = ((const V8*)bias)[idx];
      gm[j] = ((const V8*)gamma)[idx];
      bt[j] = ((const V8*)beta)[idx];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long row = row0 + r;
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = lane + 64 * j;
      if (bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[r][j][e] += (float)bs[j][e];
      }
      if (s_out) {
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = (T)v[r][j][e];
          v[r][j][e] = (float)o[e];
        }
        if (idx < nvec) ((V8*)(s_out + row * C))[idx] = o;
      }
      if (idx < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) t += v[r][j][e];
      }
    }
    sum[r] = t;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float mean = wave_sum(sum[r]) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (lane + 64 * j < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dlt = v[r][j][e] - mean;
          q += dlt * dlt;
        }
      }
    sq[r] = q;
    const float rstd = rsqrtf(wave_sum(sq[r]) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = lane + 64 * j;
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (T)((v[r][j][e] - mean) * rstd * (float)gm[j][e] + (float)bt[j][e]);
      if (idx < nvec) *(V8*)(tile + idx * CHUNK + (4 * wv + r) * 16) = o;
    }
  }
  __syncthreads();
  V8* yo = (V8*)y + (size_t)blockIdx.x * nvec * 16;
  for (int t = threadIdx.x; t < nvec * 16; t += 256) yo[t] = *(const V8*)(tile + (t >> 4) * CHUNK + (t & 15) * 16);
}

// --------------------------------------------------------------------------------------------------
// y = a + b + bias[c] over [B][C][HW]
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void add_bias_nchw_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                          const T* __restrict__ bias, T* __restrict__ y, long nvec,
                                                          int C, int HWv) {
  using V8 = typename V8T<T>::type;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    const V8 av = ((const V8*)a)[i];
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (float)av[e];
    if (b) {
      const V8 bv = ((const V8*)b)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] += (float)bv[e];
    }
    if (bias) {
      const float bc = (float)bias[(int)((i / HWv) % C)];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] += bc;
    }
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (T)t[e];
    ((V8*)y)[i] = o;
  }
}

int launched(const char* what) {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "%s launch: %s", what, hipGetErrorString(e));
}

template <typename T>
int gn_dispatch(const void* x, const float* add, const void* gamma, const void* beta, void* y, int B, int C, int HW,
                int G, float eps, int silu, hipStream_t st) {
  const long nvec = (long)(C / G) * HW / 8;
  const dim3 grid(B * G);
#define STA_GN(NT, NV, CACHED)                                                                                     \
  hipLaunchKernelGGL((gn_silu_kernel<T, NT, NV, CACHED>), grid, dim3(NT), 0, st, (const T*)x, add, (const T*)gamma, \
                     (const T*)beta, (T*)y, C, HW, G, eps, silu)
  if (nvec <= 256 * 2) STA_GN(256, 2, true);
  else if (nvec <= 256 * 8) STA_GN(256, 8, true);
  else if (nvec <= 1024 * 4) STA_GN(1024, 4, true);
  else if (nvec <= 1024 * 8) STA_GN(1024, 8, true);
  else STA_GN(1024, 1, false);
#undef STA_GN
  return launched("groupnorm_silu");
}

}  // namespace

extern "C" {

int sta_groupnorm_silu(const void* x, const float* add, const void* gamma, const void* beta, void* y, int B, int C,
                       int HW, int G, float eps, int silu, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!x || !gamma || !beta || !y) return sta_fail(STA_E_ARG, "null pointer");
  if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G || HW % 8)
    return sta_fail(STA_E_ARG, "groupnorm: B=%d C=%d HW=%d G=%d (need C %% G == 0, HW %% 8 == 0)", B, C, HW, G);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  return dtype == STA_BF16 ? gn_dispatch<__bf16>(x, add, gamma, beta, y, B, C, HW, G, eps, silu, st)
                           : gn_dispatch<_Float16>(x, add, gamma, beta, y, B, C, HW, G, eps, silu, st);
}

int sta_geglu(const void* x, void* y, long R, int D, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!x || !y) return sta_fail(STA_E_ARG, "null pointer");
  if (R <= 0 || D <= 0 || D % 8) return sta_fail(STA_E_ARG, "geglu: R=%ld D=%d (need D %% 8 == 0)", R, D);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const long nvec = R * (D / 8);
  long blocks = (nvec + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(geglu_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, st, (const __bf16*)x, (__bf16*)y, nvec, D / 8);
  else
    hipLaunchKernelGGL(geglu_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, st, (const _Float16*)x, (_Float16*)y, nvec, D / 8);
  return launched("geglu");
}

int sta_add_layernorm(const void* x, const void* f, const void* bias, const void* gamma, const void* beta, void* s,
                      void* y, long R, int C, float eps, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!x || !gamma || !beta || !y) return sta_fail(STA_E_ARG, "null pointer");
  if (R <= 0 || C <= 0 || C % 8 || C > 2048) return sta_fail(STA_E_ARG, "add_layernorm: R=%ld C=%d (need C %% 8 == 0, C <= 2048)", R, C);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const unsigned blocks = (unsigned)((R + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(add_layernorm_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)x, (const __bf16*)f,
                       (const __bf16*)bias, (const __bf16*)gamma, (const __bf16*)beta, (__bf16*)s, (__bf16*)y, R, C, eps);
  else
    hipLaunchKernelGGL(add_layernorm_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, (const _Float16*)x, (const _Float16*)f,
                       (const _Float16*)bias, (const _Float16*)gamma, (const _Float16*)beta, (_Float16*)s, (_Float16*)y, R, C, eps);
  return launched("add_layernorm");
}

int sta_add_layernorm_qfrag(const void* x, const void* f, const void* bias, const void* gamma, const void* beta, void* s,
                            void* y, long R, int C, float eps, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!x || !gamma || !beta || !y) return sta_fail(STA_E_ARG, "null pointer");
  if (R <= 0 || R % 16 || C <= 0 || C % 32 || C > 1024)
    return sta_fail(STA_E_ARG, "add_layernorm_qfrag: R=%ld C=%d (need R %% 16 == 0, C %% 32 == 0, C <= 1024)", R, C);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const unsigned blocks = (unsigned)(R / 16);
  hipStream_t st = (hipStream_t)stream;
  if (C > 512) {      // two chunks per lane
    if (dtype == STA_BF16)
      hipLaunchKernelGGL(add_layernorm_qfrag_wide_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)x, (const __bf16*)f,
                         (const __bf16*)bias, (const __bf16*)gamma, (const __bf16*)beta, (__bf16*)s, (__bf16*)y, R, C, eps);
    else
      hipLaunchKernelGGL(add_layernorm_qfrag_wide_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, (const _Float16*)x, (const _Float16*)f,
                         (const _Float16*)bias, (const _Float16*)gamma, (const _Float16*)beta, (_Float16*)s, (_Float16*)y, R, C, eps);
    return launched("add_layernorm_qfrag");
  }
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(add_layernorm_qfrag_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)x, (const __bf16*)f,
                       (const __bf16*)bias, (const __bf16*)gamma, (const __bf16*)beta, (__bf16*)s, (__bf16*)y, R, C, eps);
  else
    hipLaunchKernelGGL(add_layernorm_qfrag_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, (const _Float16*)x, (const _Float16*)f,
                       (const _Float16*)bias, (const _Float16*)gamma, (const _Float16*)beta, (_Float16*)s, (_Float16*)y, R, C, eps);
  return launched("add_layernorm_qfrag");
}

int sta_add_bias_nchw(const void* a, const void* b, const void* bias, void* y, int B, int C, int HW, int dtype,
                      void* stream) {
  g_sta_err[0] = 0;
  if (!a || !y) return sta_fail(STA_E_ARG, "null pointer");
  if (B <= 0 || C <= 0 || HW <= 0 || HW % 8) return sta_fail(STA_E_ARG, "add_bias: B=%d C=%d HW=%d (need HW %% 8 == 0)", B, C, HW);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const long nvec = (long)B * C * (HW / 8);
  long blocks = (nvec + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(add_bias_nchw_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, st, (const __bf16*)a, (const __bf16*)b,
                       (const __bf16*)bias, (__bf16*)y, nvec, C, HW / 8);
  else
    hipLaunchKernelGGL(add_bias_nchw_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, st, (const _Float16*)a, (const _Float16*)b,
                       (const _Float16*)bias, (_Float16*)y, nvec, C, HW / 8);
  return launched("add_bias_nchw");
}

static int gn_nhwc_chunks(int HW) {
  int n = HW / 8;
  if (n > 32) n = 32;
  if (n < 1) n = 1;
  return n;
}

size_t sta_groupnorm_nhwc_workspace_bytes(int B, int HW, int G) {
  if (B <= 0 || HW <= 0 || G <= 0) return 0;
  return (size_t)B * gn_nhwc_chunks(HW) * 2 * G * sizeof(float);
}

static int groupnorm_silu_nhwc_impl(const void* x, const void* xb, int Ca, const float* add, const void* gamma, const void* beta, void* y,
                                    void* workspace, int B, int C, int HW, int G, float eps, int silu, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!x || !gamma || !beta || !y || !workspace) return sta_fail(STA_E_ARG, "null pointer");
  if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || G > GN_MAXG || C % G || C % 8 || (C / G < 8 && C / G != 4) || C / 8 > GN_NT)
    return sta_fail(STA_E_ARG, "groupnorm nhwc: B=%d C=%d HW=%d G=%d (need C %% 8 == 0, C/G >= 8 or == 4, C <= %d, G <= %d)", B, C, HW,
                    G, 8 * GN_NT, GN_MAXG);
  if (xb && (Ca <= 0 || Ca >= C || Ca % 8)) return sta_fail(STA_E_ARG, "groupnorm nhwc cat: Ca=%d of C=%d (need 0 < Ca < C, Ca %% 8 == 0)", Ca, C);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const int nchunk = gn_nhwc_chunks(HW), chunk_px = (HW + nchunk - 1) / nchunk;
  const dim3 grid(nchunk, B);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == STA_BF16) {
    hipLaunchKernelGGL(gn_nhwc_stats_kernel<__bf16>, grid, dim3(GN_NT), 0, st, (const __bf16*)x, (const __bf16*)xb, Ca, add, part, C, HW, G, chunk_px);
    hipLaunchKernelGGL(gn_nhwc_apply_kernel<__bf16>, grid, dim3(GN_NT), 0, st, (const __bf16*)x, (const __bf16*)xb, Ca, add, (const float*)nullptr, (const float*)nullptr, (const __bf16*)gamma,
                       (const __bf16*)beta, part, (__bf16*)y, C, HW, G, chunk_px, eps, silu);
  } else {
    hipLaunchKernelGGL(gn_nhwc_stats_kernel<_Float16>, grid, dim3(GN_NT), 0, st, (const _Float16*)x, (const _Float16*)xb, Ca, add, part, C, HW, G, chunk_px);
    hipLaunchKernelGGL(gn_nhwc_apply_kernel<_Float16>, grid, dim3(GN_NT), 0, st, (const _Float16*)x, (const _Float16*)xb, Ca, add, (const float*)nullptr, (const float*)nullptr, (const _Float16*)gamma,
                       (const _Float16*)beta, part, (_Float16*)y, C, HW, G, chunk_px, eps, silu);
  }
  return launched("groupnorm_silu_nhwc");
}

int sta_groupnorm_silu_nhwc(const void* x, const float* add, const void* gamma, const void* beta, void* y,
                            void* workspace, int B, int C, int HW, int G, float eps, int silu, int dtype, void* stream) {
  return groupnorm_silu_nhwc_impl(x, nullptr, 0, add, gamma, beta, y, workspace, B, C, HW, G, eps, silu, dtype, stream);
}

int sta_groupnorm_silu_nhwc_cat(const void* xa, const void* xb, int Ca, const float* add, const void* gamma, const void* beta, void* y,
                                void* workspace, int B, int C, int HW, int G, float eps, int silu, int dtype, void* stream) {
  if (!xb) return sta_fail(STA_E_ARG, "null pointer");
  return groupnorm_silu_nhwc_impl(xa, xb, Ca, add, gamma, beta, y, workspace, B, C, HW, G, eps, silu, dtype, stream);
}

int sta_stats_finalize(const float* partial, float* stats, int B, int slots, int C, void* stream) {
  g_sta_err[0] = 0;
  if (!partial || !stats) return sta_fail(STA_E_ARG, "null pointer");
  if (B <= 0 || slots <= 0 || C <= 0 || C % 2) return sta_fail(STA_E_ARG, "stats_finalize: B=%d slots=%d C=%d (C even)", B, slots, C);
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((2 * C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, partial, stats, slots, C);
  return launched("stats_finalize");
}

int sta_groupnorm_silu_nhwc_cstats(const void* xa, const void* xb, int Ca, const float* stats_a, const float* stats_b, const float* add,
                                   const void* gamma, const void* beta, void* y, int B, int C, int HW, int G, float eps, int silu, int dtype,
                                   void* stream) {
  g_sta_err[0] = 0;
  if (!xa || !stats_a || !gamma || !beta || !y || (xb && !stats_b)) return sta_fail(STA_E_ARG, "null pointer");
  if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || G > GN_MAXG || C % G || C % 8 || (C / G < 8 && C / G != 4) || C / 8 > GN_NT)
    return sta_fail(STA_E_ARG, "groupnorm nhwc: B=%d C=%d HW=%d G=%d (need C %% 8 == 0, C/G >= 8 or == 4, C <= %d, G <= %d)", B, C, HW,
                    G, 8 * GN_NT, GN_MAXG);
  if (xb && (Ca <= 0 || Ca >= C || Ca % 8)) return sta_fail(STA_E_ARG, "groupnorm nhwc cat: Ca=%d of C=%d (need 0 < Ca < C, Ca %% 8 == 0)", Ca, C);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const int nchunk = gn_nhwc_chunks(HW), chunk_px = (HW + nchunk - 1) / nchunk;
  const dim3 grid(nchunk, B);
  hipStream_t st = (hipStream_t)stream;
  if (!xb) Ca = C;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(gn_nhwc_apply_kernel<__bf16>, grid, dim3(GN_NT), 0, st, (const __bf16*)xa, (const __bf16*)xb, Ca, add, stats_a, stats_b,
                       (const __bf16*)gamma, (const __bf16*)beta, (const float*)nullptr, (__bf16*)y, C, HW, G, chunk_px, eps, silu);
  else
    hipLaunchKernelGGL(gn_nhwc_apply_kernel<_Float16>, grid, dim3(GN_NT), 0, st, (const _Float16*)xa, (const _Float16*)xb, Ca, add, stats_a, stats_b,
                       (const _Float16*)gamma, (const _Float16*)beta, (const float*)nullptr, (_Float16*)y, C, HW, G, chunk_px, eps, silu);
  return launched("groupnorm_silu_nhwc_cstats");
}

int sta_add_bias_rows(const void* a, const void* b, const void* bias, void* y, long rows, int C, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!a || !y) return sta_fail(STA_E_ARG, "null pointer");
  if (rows <= 0 || C <= 0 || C % 8) return sta_fail(STA_E_ARG, "add_bias_rows: rows=%ld C=%d (need C %% 8 == 0)", rows, C);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const long nvec = rows * (C / 8);
  long blocks = (nvec + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(add_bias_rows_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, st, (const __bf16*)a, (const __bf16*)b,
                       (const __bf16*)bias, (__bf16*)y, nvec, C / 8);
  else
    hipLaunchKernelGGL(add_bias_rows_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, st, (const _Float16*)a, (const _Float16*)b,
                       (const _Float16*)bias, (_Float16*)y, nvec, C / 8);
  return launched("add_bias_rows");
}

}  // extern "C"
