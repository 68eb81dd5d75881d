// sta_xattn_dev.h — device-side building blocks shared by the cross-attention kernels of libsta_xattn.so
// (sta_xattn.hip: pack / forward / backward; sta_xattn_proj.hip: forward with the query projection inside).
// gfx950 only. See sta_xattn.hip for the MFMA operand conventions.
#ifndef STA_XATTN_DEV_H
#define STA_XATTN_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sta_xattn.h"
#include "sta_internal.h"

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
  int ntiles;            // pixel tiles per head
  int ntiles_aux;        // staged forward: contexts that fit LDS at once
  int head_major;        // 1: block b -> head b % H (= XCD b % 8 when H == 8); 0: XCD-contiguous tile ranges
  int iters;             // staged forward: pixel tiles a workgroup walks with one LDS image
  int tiles;             // staged forward: pixel tiles per head (ntiles = workgroups per head)
  int n_img;             // images in this launch (blockIdx.y); every tensor has a leading image axis
  float sl2e;            // scale * log2(e)
  float scale;
};

extern __shared__ __attribute__((aligned(16))) char smem[];

// Launches cover n_img independent images (prompts) at once: blockIdx.y selects the image and every
// pointer is advanced to that image's slice ([I][2][N][C] activations, [I][K+2] packed contexts,
// [I][N] mask bits, [I][K] weights, ...). Scalar arithmetic only.
template <typename T, int NDT>
__device__ __forceinline__ Params for_image(const Params& p, int img, size_t aux_per_img) {
  Params r = p;
  const size_t act = (size_t)2 * p.N * p.C * sizeof(T);
  r.q = (const char*)p.q + img * act;
  r.out = (char*)p.out + img * act;
  if (p.dout) r.dout = (const char*)p.dout + img * act;
  r.packed = p.packed + (size_t)img * (p.K + 2) * p.H * all_frags(NDT) * FRAG;
  r.mask = p.mask + (size_t)img * p.N;
  r.coef = p.coef + (size_t)img * p.K;
  if (p.aux) r.aux = p.aux + img * aux_per_img;
  return r;
}

// Optional in-kernel timeline (build with -DSTA_TRACE, tools/trace_fwd.py): lane 0 of every wave of
// workgroup `STA_TRACE_WG` stores s_memtime at a few points. Compiled out of the product library.
#ifdef STA_TRACE
__device__ long long* g_trace = nullptr;
// the pointer and the traced workgroup are read ONCE (STA_T_INIT); each point is then one store
#define STA_T_INIT()                                                                       \
  long long* trace_base = g_trace;                                                          \
  const bool trace_on = trace_base && blockIdx.x == (unsigned)trace_base[0] && blockIdx.y == 0 && (threadIdx.x & 63) == 0; \
  long long* trace_row = trace_base + 8 + (threadIdx.x >> 6) * 16;                          \
  const long long trace_t0 = (long long)wall_clock64();                                    \
  if (trace_on) trace_row[15] = trace_t0
#define STA_T(i)                                                                           \
  do {                                                                                     \
    if (trace_on) __builtin_nontemporal_store((long long)__builtin_readcyclecounter(), trace_row + (i)); \
  } while (0)
#define STA_T_END()                                                                       \
  do {                                                                                     \
    if (trace_on) trace_row[14] = (long long)wall_clock64();                               \
    if (trace_base && trace_base[1] && threadIdx.x == 0) {                               \
      trace_base[128 + 2 * (blockIdx.y * gridDim.x + blockIdx.x)] = trace_t0;                                         \
      trace_base[129 + 2 * (blockIdx.y * gridDim.x + blockIdx.x)] = (long long)wall_clock64();                        \
    }                                                                                      \
  } while (0)
#else
#define STA_T_INIT() do {} while (0)
#define STA_T_END() do {} while (0)
#define STA_T(i) do {} while (0)
#endif

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

// Buffer (SRD) loads: one instruction per 16-byte access — per-lane byte offset in a VGPR, the
// per-fragment offset in an SGPR/immediate — instead of a 64-bit VALU address computation per load.
// Offsets past `bytes` read as zero, which doubles as the predicate for pixels >= N.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
template <typename V8>
__device__ __forceinline__ V8 srd_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
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


// butterfly partners without LDS: v_permlane16_swap / v_permlane32_swap exchange 16- and 32-lane rows
__device__ __forceinline__ float bfly_max(float x) {
  const unsigned u = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const unsigned v = __float_as_uint(m);
  auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float bfly_sum(float x) {
  const unsigned u = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned v = __float_as_uint(m);
  auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// softmax over keys with tree-shaped (not chained) reductions; see softmax_keys for the layout
__device__ __forceinline__ float softmax_keys_fast(f32x4 (&st)[NKT], int g, int M, float sl2e) {
  float m4[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (16 * t + 4 * g + r < M) ? st[t][r] : -3.0e38f;
    m4[t] = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
  }
  float mx = fmaxf(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])), m4[4]);
  mx = bfly_max(mx);
  const float off = mx * sl2e;
  float s4[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      st[t][r] = (16 * t + 4 * g + r < M) ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][r], sl2e, -off)) : 0.f;
    s4[t] = (st[t][0] + st[t][1]) + (st[t][2] + st[t][3]);
  }
  const float l = bfly_sum(((s4[0] + s4[1]) + (s4[2] + s4[3])) + s4[4]);
  return 1.0f / l;
}

// Predicate-free softmax for the forward kernels. The scores of padded keys (key >= M) are forced to -1e30
// through the INITIAL VALUE of the S^T accumulator (last_tile_bias; with M > 64 only the last key tile has
// padded rows), so no per-key compare/select is left in the loop: 10 v_max3, the two-step butterfly, 10 packed
// fmas, 20 v_exp, and — unless the denominator comes from the ones row of the packed V^T — 10 adds and one v_rcp:
// about a quarter of the VALU instructions of softmax_keys_fast, which matters once a launch carries several
// images and the kernel is VALU-issue bound instead of latency bound (profiles/r01_kernel_variants.md).
__device__ __forceinline__ f32x4 last_tile_bias(int g, int M) {
  f32x4 b;
#pragma unroll
  for (int r = 0; r < 4; ++r) b[r] = (16 * (NKT - 1) + 4 * g + r < M) ? 0.f : -1.0e30f;
  return b;
}
// `want_sum` false: the caller takes the denominator from the ones row of the packed V^T (row d of O^T) instead
__device__ __forceinline__ float softmax_biased(f32x4 (&st)[NKT], float sl2e, bool want_sum = true) {
  // This file is compiled with -ffinite-math-only (sta/lib.py): fmaxf on raw MFMA outputs then needs no quieting
  // v_max x,x, so the maximum runs on the unscaled scores and scale*log2(e) folds into ONE packed fma per score
  // pair (exp2(s*c - max*c)); pre-scaling the scores first cost 10 more VALU per context (level 0: 53.2 -> 49.4 us).
  float ma = fmaxf(st[0][0], st[0][1]), mb = fmaxf(st[0][2], st[0][3]);
#pragma unroll
  for (int t = 1; t < NKT; ++t) {
    ma = fmaxf(fmaxf(ma, st[t][0]), st[t][1]);
    mb = fmaxf(fmaxf(mb, st[t][2]), st[t][3]);
  }
  const float off = bfly_max(fmaxf(ma, mb)) * sl2e;
  const f32x4 offv = {off, off, off, off}, sv = {sl2e, sl2e, sl2e, sl2e};
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    st[t] = __builtin_elementwise_fma(st[t], sv, -offv);
#pragma unroll
    for (int r = 0; r < 4; ++r) st[t][r] = __builtin_amdgcn_exp2f(st[t][r]);
  }
  if (!want_sum) return 0.f;                      // wave-uniform
  f32x4 acc = st[0];
#pragma unroll
  for (int t = 1; t < NKT; ++t) acc = acc + st[t];
  const float l = bfly_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
  return __builtin_amdgcn_rcpf(l);
}

// Epilogue stores, 16 bytes per lane. In the accumulator layout a lane holds 4 consecutive head-dim values
// per 16-wide tile (8 bytes as bf16/f16); 8-byte stores are store-ISSUE bound (MI355X_MICROARCH.md: row-per-lane
// dwordx2 epilogues run at ~7 B/clk/CU, dwordx4 halves the tail). One v_permlane16_swap per dword pairs the
// lane rows g and g^1: the even row ends up with head dims 16u+4g .. +7 of tile u, the odd row with
// 16(u+1)+4(g-1) .. +7 of tile u+1, so every store is a 16-byte piece of the pixel's head row.
template <typename T, int NDT>
__device__ __forceinline__ void store_row16(T* obase, const f32x4 (&a)[NDT], int g, int d) {
  typedef __attribute__((ext_vector_type(2))) T T2;
  const bool odd = g & 1;
#pragma unroll
  for (int u = 0; u < NDT; u += 2) {
    const unsigned x0 = __builtin_bit_cast(unsigned, T2{(T)a[u][0], (T)a[u][1]});
    const unsigned x1 = __builtin_bit_cast(unsigned, T2{(T)a[u][2], (T)a[u][3]});
    unsigned y0 = 0, y1 = 0;
    if (u + 1 < NDT) {
      y0 = __builtin_bit_cast(unsigned, T2{(T)a[u + 1][0], (T)a[u + 1][1]});
      y1 = __builtin_bit_cast(unsigned, T2{(T)a[u + 1][2], (T)a[u + 1][3]});
    }
    // after the swap: even rows (own tile-u pair, partner's tile-u pair); odd rows (partner's tile-u+1 pair, own)
    auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
    auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
    const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
    const int dd = odd ? 16 * (u + 1) + 4 * (g - 1) : 16 * u + 4 * g;
    if (dd < d && (!odd || u + 1 < NDT)) *(u32x4*)(obase + dd) = v;
  }
}

// Where a context's operand fragments come from: LDS (`const V8*` = wave-uniform base + lane, fragment f at f * 64 elements) or,
// for the kernels that cannot keep every context resident (sta_xattn_proj.hip, locals-from-L2 variant), the packed image in
// global memory through a buffer descriptor — one fully coalesced 1-KiB load per fragment, per-fragment offset in an SGPR.
template <typename V8>
struct SrdFrags {
  __amdgpu_buffer_rsrc_t r;
  unsigned voff;     // lane * 16
  unsigned soff;     // byte offset of the (context, head) block inside the descriptor's range (wave-uniform)
};
template <typename V8>
__device__ __forceinline__ V8 frag_at(const V8* fr, int f) { return fr[f * 64]; }
template <typename V8>
__device__ __forceinline__ V8 frag_at(const SrdFrags<V8>& fr, int f) { return srd_load16<V8>(fr.r, fr.voff, fr.soff + 1024u * (unsigned)f); }

// One context of the LDS-resident kernel for the QT pixel tiles of a wave. KIND is compile time — 0: ""
// on the uncond row (-> au), 1: global prompt on the cond row (-> ac), 2: a local prompt, ac += w (A - au) —
// so there is no per-context select/copy of the accumulators, queries or weights left in the instruction
// stream (the runtime-`c` version spent ~2/3 of its VALU slots on v_cndmask/v_mov and scalar branches).
template <typename T, int NDT, int QT, int KIND, typename FR>
__device__ __forceinline__ void attend_staged(const FR fr, const typename Tr<T>::V8 (&q)[QT][nks_of(NDT)],
                                              const f32x4 kb4, const float sl2e, const float (&w)[QT],
                                              f32x4 (&au)[QT][NDT], f32x4 (&ac)[QT][NDT], const int sumrow) {
  using V8 = typename Tr<T>::V8;
  constexpr int NKS = nks_of(NDT);
  constexpr int NKF = NKT * NKS, NVF = NPS * NDT;
  // LDS -> registers one operand side at a time, each ahead of its MFMAs (no ds_read -> wait -> mfma chains):
  // the K side for S^T first; the V side is requested once the S^T MFMAs are issued and lands under the
  // softmax VALU work, so at most one side (40 of the 76 fragment registers at d = 40) is live at a time.
  // Each fragment serves QT pixel tiles, whose independent softmax chains interleave.
  // Large head dims (JIT: NDT >= 7) fetch fragments per key tile / per head-dim tile right before their MFMAs, two
  // tiles in flight: 8*NKS + 24 fragment registers instead of 20*NKS + 12*NDT (220 at d = 160), which is what lets
  // EIGHT waves share one LDS image there (2 waves per SIMD need <= 256 registers each).
  constexpr bool JIT = NDT >= 7;
  // operands through a buffer descriptor (locals from L2): nothing orders those loads, hipcc would hoist all 25 + 30 fragments of a
  // d = 160 context to the top (107 - 410 spilled registers at 8 waves) — scheduling barriers keep the two-tiles-in-flight order
  constexpr bool PIN = JIT && !std::is_pointer<FR>::value;
  f32x4 st[QT][NKT];
  V8 va[JIT ? 1 : NVF];
  if constexpr (JIT) {
    V8 kt[2][NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) kt[0][s] = frag_at<V8>(fr, s);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      if (t + 1 < NKT) {
#pragma unroll
        for (int s = 0; s < NKS; ++s) kt[(t + 1) & 1][s] = frag_at<V8>(fr, ((t + 1) * NKS + s));
      }
      if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 acc = (t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NKS; ++s) acc = Tr<T>::mfma(kt[t & 1][s], q[qt][s], acc);
        st[qt][t] = acc;
      }
      if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    V8 ka[NKF];
#pragma unroll
    for (int f = 0; f < NKF; ++f) ka[f] = frag_at<V8>(fr, f);
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 acc = (t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NKS; ++s) acc = Tr<T>::mfma(ka[t * NKS + s], q[qt][s], acc);
        st[qt][t] = acc;
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < NVF; ++f) va[f] = frag_at<V8>(fr, (NKF + f));
  }
  float inv[QT];
  V8 pb[QT][NPS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    inv[qt] = softmax_biased(st[qt], sl2e, sumrow < 0);
    tiles_to_b<T>(st[qt], pb[qt]);
  }
  f32x4 o[QT][NDT];
  if constexpr (JIT) {
    V8 vt[2][NPS];
#pragma unroll
    for (int s = 0; s < NPS; ++s) vt[0][s] = frag_at<V8>(fr, (NKF + s * NDT));
#pragma unroll
    for (int u = 0; u < NDT; ++u) {
      if (u + 1 < NDT) {
#pragma unroll
        for (int s = 0; s < NPS; ++s) vt[(u + 1) & 1][s] = frag_at<V8>(fr, (NKF + s * NDT + u + 1));
      }
      if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NPS; ++s) acc = Tr<T>::mfma(vt[u & 1][s], pb[qt][s], acc);
        o[qt][u] = acc;
      }
      if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int u = 0; u < NDT; ++u)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NPS; ++s) acc = Tr<T>::mfma(va[s * NDT + u], pb[qt][s], acc);
        o[qt][u] = acc;
      }
  }
  if (sumrow >= 0) {   // denominator = row `sumrow` (= d % 16) of the last O^T tile, held by lane row sumrow >> 2
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float l = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) l = (r == (sumrow & 3)) ? o[qt][NDT - 1][r] : l;
      inv[qt] = __builtin_amdgcn_rcpf(__shfl(l, 16 * (sumrow >> 2) + (threadIdx.x & 15)));
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float wi = w[qt] * inv[qt];
#pragma unroll
    for (int u = 0; u < NDT; ++u) {
      if (KIND == 0) au[qt][u] = o[qt][u] * inv[qt];
      else if (KIND == 1) ac[qt][u] = o[qt][u] * inv[qt];
      else ac[qt][u] = o[qt][u] * wi + (ac[qt][u] - au[qt][u] * w[qt]);
    }
  }
}

// --------------------------------------------------------------------------------------------------
// host-side helpers shared by sta_xattn.hip (pack, forward) and sta_xattn_bwd.hip (backward)
// --------------------------------------------------------------------------------------------------
// waves per workgroup: the largest of {4,2,1} that still gives >= 256 workgroups (one per CU);
// small levels (N = 64..256) fall to 1 wave so the launch spreads over as many CUs as possible.
inline int pick_waves(int N, int heads) {
  for (int nw = 4; nw > 1; nw >>= 1) {
    const long wgs = (long)((N + 16 * nw - 1) / (16 * nw)) * heads;
    if (wgs >= 256) return nw;
  }
  return 1;
}

inline int check_shape(int N, int C, int heads, int M, int K) {
  if (N <= 0 || C <= 0 || heads <= 0 || M <= 0 || K < 0) return sta_fail(STA_E_ARG, "non-positive dimension");
  if (C % heads) return sta_fail(STA_E_ARG, "C=%d not divisible by heads=%d", C, heads);
  const int d = C / heads;
  if (d % 8 || d > STA_MAX_HEAD_DIM) return sta_fail(STA_E_UNSUP, "head dim %d unsupported (need d%%8==0, d<=%d)", d, STA_MAX_HEAD_DIM);
  if (M > STA_MAX_KEYS) return sta_fail(STA_E_UNSUP, "M=%d keys unsupported (max %d)", M, STA_MAX_KEYS);
  if (K > STA_MAX_OBJECTS) return sta_fail(STA_E_UNSUP, "K=%d objects unsupported (max %d)", K, STA_MAX_OBJECTS);
  return STA_OK;
}

}  // namespace
#endif
