// sta_xattn.hip — spatial-temporal cross-attention for MI355X (gfx950 / CDNA4), C-ABI in
// include/sta_xattn.h. Written for gfx950 only: wave64, v_mfma_f32_16x16x32_{bf16,f16},
// global_load_lds_dwordx4 (LDS-DMA), 160 KiB LDS per CU, 8 XCDs with private L2s.
//
// What it replaces (reference file:line, all under attention_optimization/stable-diffusion/):
//   ldm/modules/attention.py:175-197  CrossAttention.forward  (QK^T, softmax over 77 keys, attn.V)
//   ldm/modules/attention.py:278-294  BasicTransformerBlock._forward: the K+1 attn2 calls and the
//                                     disc-masked, coef-weighted global/local blend
//   ldm/modules/diffusionmodules/util.py:123-145  the part of CheckpointFunction.backward that
//                                     differentiates that section w.r.t. x and coef
//
// Layout of one 16x16x32 MFMA (D = A.B + C, all kernels below use only this shape):
//   lane = 16*g + c   (g = lane>>4 in 0..3, c = lane&15)
//   A operand: lane holds A[i = c][k = 8g .. 8g+7]          (8 x 16-bit, 4 VGPRs)
//   B operand: lane holds B[k = 8g .. 8g+7][j = c]
//   C/D      : lane holds D[i = 4g + r][j = c], r = 0..3     (4 x fp32)
// Every product is computed "swapped" so that the PIXEL is the MFMA column j = lane&15:
//   S^T[key][px]  = K[key][:] . Q[px][:]      A = K rows   (packed image "KQ"), B = Q  (global, 16 B/lane)
//   O^T[dcol][px] = V^T[dcol][:] . P^T[:][px]  A = V^T      (packed image "VP"), B = P  (registers)
// A lane therefore owns ONE pixel: softmax statistics need only two cross-lane steps (xor 16, 32),
// the disc mask / blend weight is a per-lane scalar, and the S^T accumulator registers are already
// in B-operand order for the PV product once the key axis of V is permuted at pack time
// (k-slot 8g+j of PV step s  <->  key 32s + 16*(j>>2) + 4g + (j&3)); no LDS round trip for P.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sta_xattn.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int NKT = 5;      // key tiles of 16 for S^T (M <= 80)
constexpr int NPS = 3;      // key steps of 32 for PV (96 slots; slots of tile 5 are zero)
constexpr int FRAG = 1024;  // bytes of one operand fragment (64 lanes x 16 B)
constexpr int MAXK = STA_MAX_OBJECTS;

template <typename T> struct Tr;
template <> struct Tr<__bf16> {
  using V8 = bf16x8;
  using V4 = bf16x4;
  static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Tr<_Float16> {
  using V8 = f16x8;
  using V4 = f16x4;
  static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// Number of fragments per (ctx, head): forward part [KQ | VP], backward part [VQ | KP].
__host__ __device__ constexpr int nks_of(int ndt) { return (ndt + 1) / 2; }
__host__ __device__ constexpr int fwd_frags(int ndt) { return NKT * nks_of(ndt) + NPS * ndt; }
__host__ __device__ constexpr int all_frags(int ndt) { return 2 * fwd_frags(ndt); }
// The backward stages both halves of a context; two of them fit the 160 KiB LDS only up to d = 96.
__host__ __device__ constexpr bool bwd_double_buffered(int ndt) {
  return 2 * all_frags(ndt) * 1024 + 16 <= 160 * 1024;
}

// Key held by k-slot (g, j) of PV step s: slots follow the S^T accumulator order.
__host__ __device__ __forceinline__ int pv_key(int s, int g, int j) {
  return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3);
}

// Block id -> logical work id so that each XCD (block b runs on XCD b % 8) owns a CONTIGUOUS range
// of logical ids: the `heads` workgroups of one pixel tile then share one L2, and the partially used
// 128-B lines of the [N][C] rows (a head touches d*2 bytes of each row) are fetched from HBM once.
// Bijective for every grid size (cdna_hip_programming.md §5, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int X = 8;
  const int q = nwg / X, r = nwg % X;
  const int xcd = bid % X, j = bid / X;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

struct Params {
  const void* q;         // [2][N][C]
  const char* packed;    // fragment image
  const uint8_t* mask;   // [K][N]
  const float* coef;     // [K]
  void* out;             // fwd: out [2][N][C];  bwd: dq [2][N][C]
  const void* dout;      // bwd only
  float* aux;            // fwd: maps or null; bwd: dcoef partials workspace
  int N, C, H, d, M, K;
  int ntiles;            // pixel tiles per head = ceil(N / (16*NW))
  float sl2e;            // scale * log2(e)
  float scale;
};

extern __shared__ __attribute__((aligned(16))) char smem[];

// Optional in-kernel timeline (build with -DSTA_TRACE, tools/trace_fwd.py): lane 0 of every wave of
// workgroup `STA_TRACE_WG` stores s_memtime at a few points. Compiled out of the product library.
#ifdef STA_TRACE
__device__ long long* g_trace = nullptr;
#define STA_T(i)                                                                          \
  do {                                                                                    \
    if (g_trace && blockIdx.x == (unsigned)g_trace[0] && (threadIdx.x & 63) == 0)          \
      g_trace[8 + (threadIdx.x >> 6) * 16 + (i)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define STA_T(i) do {} while (0)
#endif

// --------------------------------------------------------------------------------------------------
// pack: K,V [n_ctx][M][C] -> fragment image. One 64-lane block per fragment.
// --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void pack_kv_kernel(const T* __restrict__ k, const T* __restrict__ v,
                                                     T* __restrict__ packed, int n_ctx, int M, int C,
                                                     int H, int d, int ndt) {
  const int nks = nks_of(ndt);
  const int nfwd = NKT * nks + NPS * ndt;
  const int frag = blockIdx.x;            // 0 .. 2*nfwd-1
  const int ch = blockIdx.y;              // ctx * H + h
  const int ctx = ch / H, h = ch % H;
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const bool second = frag >= nfwd;       // backward half: roles of K and V swapped
  const int f = second ? frag - nfwd : frag;
  const T* qk_src = second ? v : k;       // source of the "QK-style" fragments (KQ / VQ)
  const T* pv_src = second ? k : v;       // source of the "PV-style" fragments (VP / KP)
  typename Tr<T>::V8 val;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    T x = (T)0.0f;
    if (f < NKT * nks) {  // rows = keys 16t + c, k-slots = head-dim 32s + 8g + j
      const int t = f / nks, s = f % nks;
      const int key = 16 * t + c, dd = 32 * s + 8 * g + j;
      if (key < M && dd < d) x = qk_src[((size_t)ctx * M + key) * C + h * d + dd];
    } else {              // rows = head-dim 16u + c, k-slots = permuted keys
      const int f2 = f - NKT * nks;
      const int s = f2 / ndt, u = f2 % ndt;
      const int key = pv_key(s, g, j), dd = 16 * u + c;
      if (key < M && dd < d) x = pv_src[((size_t)ctx * M + key) * C + h * d + dd];
    }
    val[j] = x;
  }
  typename Tr<T>::V8* dst =
      (typename Tr<T>::V8*)((char*)packed + ((size_t)ch * (2 * nfwd) + frag) * FRAG) + lane;
  *dst = val;
}

// --------------------------------------------------------------------------------------------------
// shared helpers
// --------------------------------------------------------------------------------------------------
// Issue the LDS-DMA copy of `nfr` consecutive fragments (1 KiB each) from global to LDS. The image is
// already in lane order, so destination = wave-uniform base + lane*16 is exactly what
// global_load_lds_dwordx4 writes. Waves take fragments round-robin.
__device__ __forceinline__ void stage_frags(const char* gsrc, char* ldst, int nfr, int wv, int nw,
                                            int lane) {
  for (int f = wv; f < nfr; f += nw) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(gsrc + (size_t)f * FRAG + lane * 16),
        (__attribute__((address_space(3))) void*)(ldst + f * FRAG), 16, 0, 0);
  }
}

__device__ __forceinline__ void wait_dma_and_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Softmax over the key axis of S^T tiles held as st[t][r] <-> key 16t + 4g + r, pixel = lane&15.
// On return st holds exp2((s - max) * sl2e) (0 for key >= M) and the return value is 1 / sum.
__device__ __forceinline__ float softmax_keys(f32x4 (&st)[NKT], int g, int M, float sl2e) {
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * t + 4 * g + r < M) mx = fmaxf(mx, st[t][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = (16 * t + 4 * g + r < M) ? __builtin_amdgcn_exp2f((st[t][r] - mx) * sl2e) : 0.f;
      st[t][r] = e;
      l += e;
    }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  return 1.0f / l;
}

// S^T accumulator tiles -> B operands of the 3 key steps of a PV-style product.
template <typename T>
__device__ __forceinline__ void tiles_to_b(const f32x4 (&st)[NKT], typename Tr<T>::V8 (&pb)[NPS]) {
#pragma unroll
  for (int s = 0; s < NPS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 2 * s + (j >> 2);
      pb[s][j] = (t < NKT) ? (T)st[t][j & 3] : (T)0.0f;
    }
}

template <typename T, int NKS>
__device__ __forceinline__ void load_b_frags(const T* base, bool valid, int g, int d,
                                             typename Tr<T>::V8 (&f)[NKS]) {
  using V8 = typename Tr<T>::V8;
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    V8 z = {};
    const int dd = 32 * s + 8 * g;
    f[s] = (valid && dd < d) ? *(const V8*)(base + dd) : z;
  }
}

// --------------------------------------------------------------------------------------------------
// forward: one wave per CONTEXT, operand fragments straight from L2, one LDS combine
// --------------------------------------------------------------------------------------------------
// Workgroup = 4 waves = one (pixel tile of 16*QT pixels, head). Wave w attends contexts w, w+4, ...
// (K = 2: wave 0 = "" on the uncond row, wave 1 = global prompt, waves 2/3 = local prompts; a local
// wave whose tile misses its disc has nothing to do). The K+2 attentions of a tile are independent
// until the blend, so they run side by side instead of one after another, and there is no K/V
// staging at all: the packed image is already in MFMA A-operand order, so every fragment is one
// fully coalesced 1-KiB global_load_dwordx4 per wave that hits L2 (the image of a block is 0.6-1.8
// MB and is shared by every workgroup). Each fragment is reused for QT pixel tiles (B operands),
// which is what keeps L2 traffic at the level an LDS-staged design would have.
// The only cross-wave step is the blend: every wave leaves its fp32 partial in LDS, one barrier, then
// all 256 threads combine  out1 = sum(partials) - (sum_i coef_i mask_i) * A_u  and write both rows
// with 16-byte stores.
constexpr int NSLOT = 5;   // LDS slots: 0 = A_u (row 0), 1 + w = row-1 partial of wave w

template <typename T, int NDT, int QT>
__global__ __launch_bounds__(256) void xattn_fwd_kernel(const Params p) {
  using V8 = typename Tr<T>::V8;
  constexpr int NKS = nks_of(NDT);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = L / p.H, h = L % p.H;
  const int N = p.N, C = p.C, d = p.d, K = p.K;
  const int px0 = tile * 16 * QT;
  const int DP = d + 4;                           // fp32 row stride of a slot: conflict-free b128 writes
  const int slot_floats = 16 * QT * DP;
  float* slots = (float*)smem;
  STA_T(0);

  // which discs touch this tile (every wave computes the same answer: no LDS flag, no barrier).
  // The K byte loads are issued back to back (clamped index instead of a branch per object) so they
  // cost one L2 round trip, and they are the oldest loads in flight: waves 0/1, which always have
  // work, do not wait for them before requesting Q and their K fragments.
  unsigned tile_bits = 0;
  {
    const int pxl = px0 + lane;
    const bool in_tile = lane < 16 * QT && pxl < N && K > 0;
    uint8_t mb[MAXK];
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
      const int ii = i < K ? i : (K > 0 ? K - 1 : 0);
      mb[i] = in_tile ? p.mask[(size_t)ii * N + pxl] : (uint8_t)0;
    }
#pragma unroll
    for (int i = 0; i < MAXK; ++i)
      if (i < K && __any(mb[i] != 0)) tile_bits |= 1u << i;
    if (p.aux) tile_bits = (1u << K) - 1u;        // parity mode: every map is wanted
  }

  const size_t ctx_stride = (size_t)p.H * all_frags(NDT) * FRAG;
  const char* img_h = p.packed + (size_t)h * all_frags(NDT) * FRAG;

  f32x4 part[QT][NDT];                            // this wave's row-1 partial (wave 0, ctx 0: A_u)
  bool have_part = false;
  for (int c = wv; c < K + 2; c += 4) {
    if (c >= 2 && !((tile_bits >> (c - 2)) & 1u)) continue;
    const int row = c == 0 ? 0 : 1;
    const V8* frag = (const V8*)(img_h + (size_t)c * ctx_stride) + lane;   // fragment f at frag[f*64]

    // B operands: this wave's row of Q for its QT pixel tiles (16 B per lane, d-offset 32s + 8g)
    V8 qf[QT][NKS];
    bool valid[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int px = px0 + 16 * qt + c16;
      valid[qt] = px < N;
      load_b_frags<T, NKS>((const T*)p.q + ((size_t)row * N + px) * C + h * d, valid[qt], g, d, qf[qt]);
    }

    // All K-side fragments are requested before the first MFMA: left to itself hipcc emits
    // load -> s_waitcnt vmcnt(0) -> mfma per fragment, i.e. one L2 round trip per MFMA (measured:
    // 13 us for 55 fragments). With the loads issued back to back the wave pays the L2 latency once.
    STA_T(1);
    V8 ka[NKT * NKS];
#pragma unroll
    for (int f = 0; f < NKT * NKS; ++f) ka[f] = frag[f * 64];
    __builtin_amdgcn_sched_barrier(0);
    STA_T(2);

    // S^T = K Q^T : every A fragment is used for QT pixel tiles
    f32x4 st[QT][NKT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int t = 0; t < NKT; ++t) st[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) st[qt][t] = Tr<T>::mfma(ka[t * NKS + s], qf[qt][s], st[qt][t]);

    // V-side fragments are requested now; the softmax below runs while they are in flight
    __builtin_amdgcn_sched_barrier(0);
    STA_T(3);
    V8 va[NPS * NDT];
#pragma unroll
    for (int f = 0; f < NPS * NDT; ++f) va[f] = frag[(NKT * NKS + f) * 64];
    __builtin_amdgcn_sched_barrier(0);
    STA_T(4);

    float inv[QT];
    V8 pb[QT][NPS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      inv[qt] = softmax_keys(st[qt], g, p.M, p.sl2e);
      if (p.aux && valid[qt]) {
        float* mrow = p.aux + (((size_t)c * p.H + h) * N + (px0 + 16 * qt + c16)) * p.M;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * t + 4 * g + r;
            if (key < p.M) mrow[key] = st[qt][t][r] * inv[qt];
          }
      }
      tiles_to_b<T>(st[qt], pb[qt]);
    }

    STA_T(5);
    // O^T = V^T P^T, then this context's share of the blend
    float wc[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      wc[qt] = inv[qt];
      if (c >= 2) {
        const int px = px0 + 16 * qt + c16;
        const bool m = valid[qt] && p.mask[(size_t)(c - 2) * N + px] != 0;
        wc[qt] = m ? inv[qt] * p.coef[c - 2] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < NDT; ++u) {
      f32x4 acc[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) acc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NPS; ++s)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc[qt] = Tr<T>::mfma(va[s * NDT + u], pb[qt][s], acc[qt]);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
        part[qt][u] = (have_part && c != 0) ? part[qt][u] + acc[qt] * wc[qt] : acc[qt] * wc[qt];
    }
    if (c == 0) {
      // A_u goes to slot 0 right away; wave 0 may go on with local contexts 4, 8 (row-1 partial)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int u = 0; u < NDT; ++u)
          if (16 * u + 4 * g < d) *(f32x4*)(slots + (16 * qt + c16) * DP + 16 * u + 4 * g) = part[qt][u];
    } else {
      have_part = true;
    }
  }
  if (have_part) {
    float* sl = slots + (1 + wv) * slot_floats;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int u = 0; u < NDT; ++u)
        if (16 * u + 4 * g < d) *(f32x4*)(sl + (16 * qt + c16) * DP + 16 * u + 4 * g) = part[qt][u];
  }
  STA_T(6);
  // which slots hold a row-1 partial: wave w if any of its contexts w, w+4, ... (>= 1) was active
  unsigned slot_bits = 0;
  for (int c = 1; c < K + 2; ++c)
    if (c < 2 || ((tile_bits >> (c - 2)) & 1u)) slot_bits |= 1u << (c & 3);
  __syncthreads();
  STA_T(7);

  // combine + store: thread -> (pixel, 8-channel chunk); 16-byte stores to both rows
  const int chunks = d >> 3;
  for (int it = threadIdx.x; it < 16 * QT * chunks; it += 256) {
    const int pl = it / chunks, ch = it - pl * chunks;
    const int px = px0 + pl;
    if (px >= N) continue;
    float wsum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXK; ++i)
      if (i < K && p.mask[(size_t)i * N + px] != 0) wsum += p.coef[i];
    const float* s0 = slots + pl * DP + 8 * ch;
    float au[8], o1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      au[j] = s0[j];
      o1[j] = -wsum * au[j];
    }
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if ((slot_bits >> w) & 1u) {
        const float* sw = slots + (1 + w) * slot_floats + pl * DP + 8 * ch;
#pragma unroll
        for (int j = 0; j < 8; ++j) o1[j] += sw[j];
      }
    V8 r0, r1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      r0[j] = (T)au[j];
      r1[j] = (T)o1[j];
    }
    T* ob = (T*)p.out + (size_t)px * C + h * d + 8 * ch;
    *(V8*)ob = r0;
    *(V8*)(ob + (size_t)N * C) = r1;
  }
  STA_T(8);
}

// --------------------------------------------------------------------------------------------------
// backward (dq, dcoef)
// --------------------------------------------------------------------------------------------------
// For one context with probabilities P (normalised), upstream gradient G on A = P V:
//   dP = G V^T ; delta = sum_key P dP ; dS = P (dP - delta) ; dQ = scale * dS K
// Returns (optionally) A in o[] for the dcoef dot product and accumulates dQ^T tiles into dq[].
template <typename T, int NDT, bool WANT_A>
__device__ __forceinline__ void attend_bwd(const char* buf, const typename Tr<T>::V8 (&qf)[nks_of(NDT)],
                                           const typename Tr<T>::V8 (&gf)[nks_of(NDT)], float gscale,
                                           f32x4 (&o)[NDT], f32x4 (&dq)[NDT], int lane, int g, int M,
                                           float sl2e) {
  using V8 = typename Tr<T>::V8;
  constexpr int NKS = nks_of(NDT);
  constexpr int NFWD = fwd_frags(NDT);
  const V8* frag = (const V8*)buf + lane;
  f32x4 st[NKT], dp[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      st[t] = Tr<T>::mfma(frag[(t * NKS + s) * 64], qf[s], st[t]);
      dp[t] = Tr<T>::mfma(frag[(NFWD + t * NKS + s) * 64], gf[s], dp[t]);  // VQ . G^T
    }
  }
  const float inv = softmax_keys(st, g, M, sl2e);
  V8 pb[NPS];
  if (WANT_A) {
    tiles_to_b<T>(st, pb);
#pragma unroll
    for (int u = 0; u < NDT; ++u) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NPS; ++s) acc = Tr<T>::mfma(frag[(NKT * NKS + s * NDT + u) * 64], pb[s], acc);
      o[u] = acc * inv;
    }
  }
  // delta and dS (in place in st). Padded keys have st == 0, so they contribute nothing.
  float delta = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      st[t][r] *= inv;
      delta += st[t][r] * dp[t][r];
    }
  delta += __shfl_xor(delta, 16);
  delta += __shfl_xor(delta, 32);
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) st[t][r] = st[t][r] * (dp[t][r] - delta) * gscale;
  tiles_to_b<T>(st, pb);
#pragma unroll
  for (int u = 0; u < NDT; ++u)
#pragma unroll
    for (int s = 0; s < NPS; ++s)
      dq[u] = Tr<T>::mfma(frag[(NFWD + NKT * NKS + s * NDT + u) * 64], pb[s], dq[u]);  // KP . dS^T
}

template <typename T, int NDT>
__global__ __launch_bounds__(256) void xattn_bwd_kernel(const Params p) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  constexpr int NKS = nks_of(NDT);
  constexpr int NALL = all_frags(NDT);
  constexpr int CB = NALL * FRAG;
  constexpr bool DB = bwd_double_buffered(NDT);  // forward+backward images of two contexts in LDS?
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = L / p.H, h = L % p.H;
  const int px = (tile * nw + wv) * 16 + c16;
  const bool valid = px < p.N;
  const int N = p.N, C = p.C, d = p.d, K = p.K;
  unsigned* flags = (unsigned*)(smem + (DB ? 2 : 1) * CB);

  float w[MAXK], dc[MAXK];
  float wsum = 0.f;
  unsigned mybits = 0;
#pragma unroll
  for (int i = 0; i < MAXK; ++i) {
    w[i] = 0.f;
    dc[i] = 0.f;
    if (i < K) {
      const bool m = valid && p.mask[(size_t)i * N + px] != 0;
      w[i] = m ? p.coef[i] : 0.f;
      wsum += w[i];
      if (__any(m)) mybits |= 1u << i;
    }
  }
  if (threadIdx.x == 0) flags[0] = 0;
  __syncthreads();
  if (lane == 0 && mybits) atomicOr(flags, mybits);
  __syncthreads();
  const unsigned wgbits = flags[0];

  const size_t row1 = (size_t)N * C;
  const T* qbase = (const T*)p.q + (size_t)px * C + h * d;
  const T* gbase = (const T*)p.dout + (size_t)px * C + h * d;
  V8 qf[NKS], gf[NKS], g1[NKS];
  // row 0: upstream of A_u is dO0 - (sum_i coef_i mask_i) dO1
  load_b_frags<T, NKS>(qbase, valid, g, d, qf);
  load_b_frags<T, NKS>(gbase, valid, g, d, gf);
  load_b_frags<T, NKS>(gbase + row1, valid, g, d, g1);
#pragma unroll
  for (int s = 0; s < NKS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) gf[s][j] = (T)((float)gf[s][j] - wsum * (float)g1[s][j]);
  // dO1 in accumulator (O^T) order for the dcoef dot products
  f32x4 g1t[NDT];
#pragma unroll
  for (int u = 0; u < NDT; ++u) {
    const int dd = 16 * u + 4 * g;
    g1t[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (valid && dd < d) {
      const V4 t4 = *(const V4*)(gbase + row1 + dd);
#pragma unroll
      for (int r = 0; r < 4; ++r) g1t[u][r] = (float)t4[r];
    }
  }

  const size_t ctx_stride = (size_t)p.H * NALL * FRAG;
  const char* img_h = p.packed + (size_t)h * NALL * FRAG;
  T* dqbase = (T*)p.out + (size_t)px * C + h * d;

  f32x4 au[NDT], o[NDT], dq[NDT];
#pragma unroll
  for (int u = 0; u < NDT; ++u) dq[u] = f32x4{0.f, 0.f, 0.f, 0.f};

  int c = 0, b = 0;
  stage_frags(img_h, smem, NALL, wv, nw, lane);
  while (c >= 0) {
    int cn = -1;
    for (int cc = c + 1; cc < K + 2; ++cc)
      if (cc < 2 || ((wgbits >> (cc - 2)) & 1u)) { cn = cc; break; }
    wait_dma_and_sync();
    if (DB && cn >= 0) stage_frags(img_h + cn * ctx_stride, smem + (b ^ 1) * CB, NALL, wv, nw, lane);
    const char* buf = smem + b * CB;
    if (c == 0) {
      if (mybits)
        attend_bwd<T, NDT, true>(buf, qf, gf, p.scale, au, dq, lane, g, p.M, p.sl2e);
      else
        attend_bwd<T, NDT, false>(buf, qf, gf, p.scale, au, dq, lane, g, p.M, p.sl2e);
      // dq row 0 is complete: store it and switch to row 1 operands
      if (valid) {
#pragma unroll
        for (int u = 0; u < NDT; ++u) {
          const int dd = 16 * u + 4 * g;
          if (dd < d) {
            V4 r0;
#pragma unroll
            for (int r = 0; r < 4; ++r) r0[r] = (T)dq[u][r];
            *(V4*)(dqbase + dd) = r0;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NDT; ++u) dq[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      load_b_frags<T, NKS>(qbase + row1, valid, g, d, qf);
    } else if (c == 1) {
      attend_bwd<T, NDT, false>(buf, qf, g1, p.scale, o, dq, lane, g, p.M, p.sl2e);
    } else if ((mybits >> (c - 2)) & 1u) {
      float wc = 0.f;
#pragma unroll
      for (int i = 0; i < MAXK; ++i) wc = (i == c - 2) ? w[i] : wc;
      // G_i = w_i dO1: linear, so feed dO1 and scale dS by w_i (exact in fp32, no re-rounding)
      attend_bwd<T, NDT, true>(buf, qf, g1, p.scale * wc, o, dq, lane, g, p.M, p.sl2e);
      // dcoef_i += mask_i(px) * sum_d dO1 (A_i - A_u)
      float part = 0.f;
#pragma unroll
      for (int u = 0; u < NDT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) part += g1t[u][r] * (o[u][r] - au[u][r]);
      const bool inside = valid && p.mask[(size_t)(c - 2) * N + px] != 0;
      part = inside ? part : 0.f;
#pragma unroll
      for (int i = 0; i < MAXK; ++i) dc[i] += (i == c - 2) ? part : 0.f;
    }
    if (!DB && cn >= 0) {  // single buffer: refill only after every wave has finished reading it
      __syncthreads();
      stage_frags(img_h + cn * ctx_stride, smem, NALL, wv, nw, lane);
    }
    c = cn;
    if (DB) b ^= 1;
  }

  if (valid) {
#pragma unroll
    for (int u = 0; u < NDT; ++u) {
      const int dd = 16 * u + 4 * g;
      if (dd < d) {
        V4 r1;
#pragma unroll
        for (int r = 0; r < 4; ++r) r1[r] = (T)dq[u][r];
        *(V4*)(dqbase + row1 + dd) = r1;
      }
    }
  }
  // per-wave dcoef partials -> workspace [K][gridDim.x * nw] (fixed slot per wave: deterministic)
#pragma unroll
  for (int i = 0; i < MAXK; ++i) {
    if (i < K) {
      float v = dc[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) p.aux[(size_t)i * gridDim.x * nw + (size_t)blockIdx.x * nw + wv] = v;
    }
  }
}

// Sum the per-wave partials in a fixed order: one block of 256 threads per object.
__global__ __launch_bounds__(256) void dcoef_reduce_kernel(const float* __restrict__ part, float* dcoef,
                                                           int n) {
  __shared__ float sm[256];
  const float* src = part + (size_t)blockIdx.x * n;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += src[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dcoef[blockIdx.x] = sm[0];
}

// --------------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------------
thread_local char g_err[256] = "";

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

// waves per workgroup: the largest of {4,2,1} that still gives >= 256 workgroups (one per CU);
// small levels (N = 64..256) fall to 1 wave so the launch spreads over as many CUs as possible.
int pick_waves(int N, int heads) {
  for (int nw = 4; nw > 1; nw >>= 1) {
    const long wgs = (long)((N + 16 * nw - 1) / (16 * nw)) * heads;
    if (wgs >= 256) return nw;
  }
  return 1;
}

int check_shape(int N, int C, int heads, int M, int K) {
  if (N <= 0 || C <= 0 || heads <= 0 || M <= 0 || K < 0) return fail(STA_E_ARG, "non-positive dimension");
  if (C % heads) return fail(STA_E_ARG, "C=%d not divisible by heads=%d", C, heads);
  const int d = C / heads;
  if (d % 8 || d > STA_MAX_HEAD_DIM) return fail(STA_E_UNSUP, "head dim %d unsupported (need d%%8==0, d<=%d)", d, STA_MAX_HEAD_DIM);
  if (M > STA_MAX_KEYS) return fail(STA_E_UNSUP, "M=%d keys unsupported (max %d)", M, STA_MAX_KEYS);
  if (K > STA_MAX_OBJECTS) return fail(STA_E_UNSUP, "K=%d objects unsupported (max %d)", K, STA_MAX_OBJECTS);
  return STA_OK;
}

// Pixel tiles per wave: each K/V fragment read from L2 is reused QT times, so larger QT cuts L2
// traffic; smaller QT gives more workgroups. Take the largest QT the register budget of the head
// dim allows that still yields >= 512 workgroups (2 per CU), else 1.
int pick_qt(int N, int heads, int ndt) {
  int cap = ndt <= 3 ? 4 : (ndt <= 6 ? 2 : 1);
  if (const char* e = getenv("STA_FWD_QT")) {  // tuning knob (tools/kernel_bench.py); not used in production
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4) return v < cap ? v : cap;
  }
  for (int qt = cap; qt > 1; qt >>= 1)
    if ((long)((N + 16 * qt - 1) / (16 * qt)) * heads >= 512) return qt;
  return 1;
}

template <typename T, int NDT, int QT>
int launch_fwd(const Params& p0, hipStream_t st) {
  Params p = p0;
  p.ntiles = (p.N + 16 * QT - 1) / (16 * QT);
  const int lds = NSLOT * 16 * QT * (p.d + 4) * (int)sizeof(float);
  static bool attr_set = false;  // benign race: idempotent
  if (!attr_set) {
    constexpr int lds_max = NSLOT * 16 * QT * (16 * NDT + 4) * (int)sizeof(float);
    if (hipFuncSetAttribute((const void*)xattn_fwd_kernel<T, NDT, QT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess)
      return fail(STA_E_LAUNCH, "hipFuncSetAttribute(fwd) failed");
    attr_set = true;
  }
  hipLaunchKernelGGL((xattn_fwd_kernel<T, NDT, QT>), dim3(p.ntiles * p.H), dim3(256), lds, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : fail(STA_E_LAUNCH, "fwd launch: %s", hipGetErrorString(e));
}

template <typename T, int NDT>
int launch_fwd_qt(const Params& p, int qt, hipStream_t st) {
  if constexpr (NDT <= 3) { if (qt == 4) return launch_fwd<T, NDT, 4>(p, st); }
  if constexpr (NDT <= 6) { if (qt >= 2) return launch_fwd<T, NDT, 2>(p, st); }
  return launch_fwd<T, NDT, 1>(p, st);
}

template <typename T, int NDT>
int launch_bwd(const Params& p, int nw, float* dcoef, hipStream_t st) {
  constexpr int lds = (bwd_double_buffered(NDT) ? 2 : 1) * all_frags(NDT) * FRAG + 16;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)xattn_bwd_kernel<T, NDT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return fail(STA_E_LAUNCH, "hipFuncSetAttribute(bwd) failed");
    attr_set = true;
  }
  const int nwg = p.ntiles * p.H;
  hipLaunchKernelGGL((xattn_bwd_kernel<T, NDT>), dim3(nwg), dim3(64 * nw), lds, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(STA_E_LAUNCH, "bwd launch: %s", hipGetErrorString(e));
  if (p.K > 0) {
    hipLaunchKernelGGL(dcoef_reduce_kernel, dim3(p.K), dim3(256), 0, st, p.aux, dcoef, nwg * nw);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(STA_E_LAUNCH, "dcoef reduce launch: %s", hipGetErrorString(e));
  }
  return STA_OK;
}

template <typename T>
int dispatch_fwd(const Params& p, hipStream_t st) {
  const int ndt = (p.d + 15) / 16;
  const int qt = pick_qt(p.N, p.H, ndt);
  switch (ndt) {
    case 1: return launch_fwd_qt<T, 1>(p, qt, st);
    case 2: return launch_fwd_qt<T, 2>(p, qt, st);
    case 3: return launch_fwd_qt<T, 3>(p, qt, st);
    case 4: return launch_fwd_qt<T, 4>(p, qt, st);
    case 5: return launch_fwd_qt<T, 5>(p, qt, st);
    case 6: return launch_fwd_qt<T, 6>(p, qt, st);
    case 7: return launch_fwd_qt<T, 7>(p, qt, st);
    case 8: return launch_fwd_qt<T, 8>(p, qt, st);
    case 9: return launch_fwd_qt<T, 9>(p, qt, st);
    case 10: return launch_fwd_qt<T, 10>(p, qt, st);
  }
  return fail(STA_E_UNSUP, "head dim %d unsupported", p.d);
}

template <typename T>
int dispatch_bwd(const Params& p, int nw, float* dcoef, hipStream_t st) {
  switch ((p.d + 15) / 16) {
    case 1: return launch_bwd<T, 1>(p, nw, dcoef, st);
    case 2: return launch_bwd<T, 2>(p, nw, dcoef, st);
    case 3: return launch_bwd<T, 3>(p, nw, dcoef, st);
    case 4: return launch_bwd<T, 4>(p, nw, dcoef, st);
    case 5: return launch_bwd<T, 5>(p, nw, dcoef, st);
    case 6: return launch_bwd<T, 6>(p, nw, dcoef, st);
    case 7: return launch_bwd<T, 7>(p, nw, dcoef, st);
    case 8: return launch_bwd<T, 8>(p, nw, dcoef, st);
    case 9: return launch_bwd<T, 9>(p, nw, dcoef, st);
    case 10: return launch_bwd<T, 10>(p, nw, dcoef, st);
  }
  return fail(STA_E_UNSUP, "head dim %d unsupported", p.d);
}

}  // namespace

extern "C" {

#ifdef STA_TRACE
// trace build only: buf[0] = workgroup to trace, buf[8 + wave*16 + i] = s_memtime at point i
int sta_debug_set_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

int sta_version(void) { return STA_VERSION; }

const char* sta_last_error(void) { return g_err; }

size_t sta_xattn_packed_kv_bytes(int n_ctx, int heads, int d) {
  if (n_ctx <= 0 || heads <= 0 || d <= 0 || d % 8 || d > STA_MAX_HEAD_DIM) return 0;
  return (size_t)n_ctx * heads * all_frags((d + 15) / 16) * FRAG;
}

int sta_xattn_pack_kv(const void* k, const void* v, void* packed, int n_ctx, int M, int C, int heads,
                      int dtype, void* stream) {
  g_err[0] = 0;
  if (!k || !v || !packed) return fail(STA_E_ARG, "null pointer");
  if (n_ctx <= 0) return fail(STA_E_ARG, "n_ctx=%d", n_ctx);
  if (int rc = check_shape(16, C, heads, M, 0)) return rc;
  if (dtype != STA_BF16 && dtype != STA_F16) return fail(STA_E_UNSUP, "dtype %d", dtype);
  const int d = C / heads, ndt = (d + 15) / 16;
  const dim3 grid(all_frags(ndt), n_ctx * heads);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(pack_kv_kernel<__bf16>, grid, dim3(64), 0, st, (const __bf16*)k, (const __bf16*)v,
                       (__bf16*)packed, n_ctx, M, C, heads, d, ndt);
  else
    hipLaunchKernelGGL(pack_kv_kernel<_Float16>, grid, dim3(64), 0, st, (const _Float16*)k,
                       (const _Float16*)v, (_Float16*)packed, n_ctx, M, C, heads, d, ndt);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : fail(STA_E_LAUNCH, "pack launch: %s", hipGetErrorString(e));
}

int sta_xattn_fwd(const void* q, const void* packed, const uint8_t* mask, const float* coef, void* out,
                  float* maps, int N, int C, int heads, int M, int K, float scale, int dtype,
                  void* stream) {
  g_err[0] = 0;
  if (!q || !packed || !out) return fail(STA_E_ARG, "null pointer");
  if (int rc = check_shape(N, C, heads, M, K)) return rc;
  if (K > 0 && (!mask || !coef)) return fail(STA_E_ARG, "mask/coef required when K > 0");
  if (dtype != STA_BF16 && dtype != STA_F16) return fail(STA_E_UNSUP, "dtype %d", dtype);
  Params p{};
  p.q = q; p.packed = (const char*)packed; p.mask = mask; p.coef = coef; p.out = out; p.dout = nullptr;
  p.aux = maps; p.N = N; p.C = C; p.H = heads; p.d = C / heads; p.M = M; p.K = K;
  p.scale = scale; p.sl2e = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  return dtype == STA_BF16 ? dispatch_fwd<__bf16>(p, st) : dispatch_fwd<_Float16>(p, st);
}

size_t sta_xattn_bwd_workspace_bytes(int N, int heads, int K) {
  if (N <= 0 || heads <= 0 || K <= 0) return 16;
  return (size_t)K * ((N + 15) / 16 + 4) * heads * sizeof(float);
}

int sta_xattn_bwd(const void* q, const void* packed, const uint8_t* mask, const float* coef,
                  const void* dout, void* dq, float* dcoef, void* workspace, int N, int C, int heads,
                  int M, int K, float scale, int dtype, void* stream) {
  g_err[0] = 0;
  if (!q || !packed || !dout || !dq) return fail(STA_E_ARG, "null pointer");
  if (int rc = check_shape(N, C, heads, M, K)) return rc;
  if (K > 0 && (!mask || !coef || !dcoef || !workspace)) return fail(STA_E_ARG, "mask/coef/dcoef/workspace required when K > 0");
  if (dtype != STA_BF16 && dtype != STA_F16) return fail(STA_E_UNSUP, "dtype %d", dtype);
  const int nw = pick_waves(N, heads);
  Params p{};
  p.q = q; p.packed = (const char*)packed; p.mask = mask; p.coef = coef; p.out = dq; p.dout = dout;
  p.aux = (float*)workspace; p.N = N; p.C = C; p.H = heads; p.d = C / heads; p.M = M; p.K = K;
  p.ntiles = (N + 16 * nw - 1) / (16 * nw);
  p.scale = scale; p.sl2e = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  return dtype == STA_BF16 ? dispatch_bwd<__bf16>(p, nw, dcoef, st) : dispatch_bwd<_Float16>(p, nw, dcoef, st);
}

}  // extern "C"
