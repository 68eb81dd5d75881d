// sta_xattn_proj2.hip — the projection-fused forward for HEAD PAIRS (d = 40: SD-v1 level 0, K <= 2).
//
// Why a second kernel: sta_xattn_proj.hip is bound by the L2 -> L1 path — the y rows of a pixel tile are read once per
// HEAD (8x the activation bytes, 671 MB per 16-image launch, arriving at 7.6 TB/s: profiles/r02_proj_fusion.md). One
// workgroup can only serve two heads from one y read if BOTH heads' operands fit the CU's 160 KiB of LDS, and in MFMA
// fragment order (keys padded 77 -> 80/96, head dim 40 -> 48/64) they do not: 2 x (30 + 4 x 19) = 212 KiB. Stored
// COMPACTLY they do:
//   Wq of the pair   80 output columns = 5 column tiles exactly (no padding)         5 x nkc KiB  = 50 KiB at C = 320
//   K  per (ctx, head)   [80 keys][40 dims] row-major, keys 77..79 zero                 6400 B
//   V^T per (ctx, head)  [41 rows = 40 dims + a row of ones][88 key slots], zero padded    7216 B
//   -> 13.5 KiB per (ctx, head), 108 KiB for 4 contexts x 2 heads; with the slack exactly the 160 KiB of a CU.
// The K / V^T operand fragments are then assembled from two 8-byte LDS reads each (k-slots 8g .. 8g+3 and 8g+4 .. 8g+7
// of a 16x16x32 MFMA are two runs of 4 consecutive dims / keys) instead of one 16-byte read of a pre-permuted
// fragment. Slots that belong to no dim of the head read whatever finite bytes lie there; the OTHER operand (q, built
// in registers) carries the zeros.
//
// The 80 projected columns of a pixel tile split as: head A = pair dims 0..39 = tiles 0, 1 and rows 0..7 of tile 2;
// head B = pair dims 40..79 = rows 8..15 of tile 2 and tiles 3, 4. S^T k-slots (step s, half hf) take pair tile
// tX0 + 2s + hf (tA0 = 0, tB0 = 2), i.e. head dims 16 (2s + hf) + 4g + r for A and 16 (2s + hf) + 4g + r - 8 for B.
//
// Reference replaced: ldm/modules/attention.py:178 (to_q), :175-197 (the K+2 attentions), :278-294 (masked blend) —
// identical arithmetic to sta_xattn_fwd_proj; reached through the same C-ABI call (sta_xattn_proj.hip dispatches here).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sta_xattn.h"
#include "sta_internal.h"
#include "sta_xattn_dev.h"
#include "sta_xattn_proj2.h"

namespace {

using namespace sta_pair;

// K, V [n_ctx][M][C] -> compact image [ctx][pair][head in pair][K block | V^T block], BLK bytes each.
template <typename T>
__global__ __launch_bounds__(256) void pack_kv_pair_kernel(const T* __restrict__ k, const T* __restrict__ v,
                                                           T* __restrict__ packed, int M, int C, int H) {
  const int ch = blockIdx.x;                  // ctx * H + h
  const int ctx = ch / H, h = ch % H;
  T* blk = (T*)((char*)packed + ((size_t)ctx * H + h) * BLK);       // [ctx][pair][hp] with h = 2 pair + hp
  for (int i = threadIdx.x; i < BLK / 2; i += blockDim.x) {
    T x = (T)0.0f;
    if (i < KROWS * D) {
      const int key = i / D, dd = i % D;
      if (key < M) x = k[((size_t)ctx * M + key) * C + h * D + dd];
    } else if (i < KROWS * D + VROWS * VSLOTS) {
      const int j = i - KROWS * D;
      const int row = j / VSLOTS, key = j % VSLOTS;
      if (key < M) x = row < D ? v[((size_t)ctx * M + key) * C + h * D + row] : (T)1.0f;     // row D: ones (softmax denominator)
    }
    blk[i] = x;
  }
}

struct P2 {
  const void* y;
  const char* wq;        // pair fragments: [pair][nkc][NT]
  const char* kv;        // compact image: [I][K+2][pairs][2][BLK]
  const uint8_t* mask;
  const float* coef;
  void* out;
  int N, C, H, M, K, nkc, W, tiles, iters;
  float sl2e;
};

typedef __attribute__((ext_vector_type(4))) unsigned short us4;      // 8 bytes of LDS: four 16-bit values

// two 8-byte LDS reads -> one A operand (k-slots 0..3 | 4..7 of the lane)
template <typename T>
__device__ __forceinline__ typename Tr<T>::V8 ld_pair(const char* lo, const char* hi) {
  const us4 a = *(const us4*)lo, b = *(const us4*)hi;
  typedef __attribute__((ext_vector_type(8))) unsigned short us8;
  const us8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(typename Tr<T>::V8, r);
}
template <typename T>
__device__ __forceinline__ typename Tr<T>::V8 ld_half(const char* lo) {     // k-slots 4..7 are zeros
  const us4 a = *(const us4*)lo;
  typedef __attribute__((ext_vector_type(8))) unsigned short us8;
  const us8 r = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
  return __builtin_bit_cast(typename Tr<T>::V8, r);
}

// One context of one head from the compact image. kb: this lane's K base (block + c*2D + 8g, minus 16 for head B);
// vb: its V^T base (block + K bytes + c*VS + 8g). KIND as in attend_staged.
struct NoFiller { __device__ __forceinline__ void operator()() const {} };

// `filler` is issued between the S^T MFMAs and the softmax, in the same scheduling region as the V^T fragment reads and
// the softmax VALU chain: the software-pipelined kernel passes a few k-steps of the NEXT tile's projection there, whose
// MFMAs the scheduler can then place under this context's vector work.
template <typename T, int KIND, int HB, typename F = NoFiller>
__device__ __forceinline__ void attend_compact(const char* kb, const char* vb, const typename Tr<T>::V8 (&q)[2], const f32x4 kb4,
                                               const float sl2e, const float w, f32x4 (&au)[3], f32x4 (&ac)[3], F filler = F()) {
  using V8 = typename Tr<T>::V8;
  // S^T = K Q^T: 5 key tiles x (step 0: both halves, step 1: first half only)
  f32x4 st[1][NKT];
  {
    V8 k0[NKT], k1[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const char* row = kb + t * 16 * 2 * D;
      k0[t] = ld_pair<T>(row, row + 32);                       // dims 4g.. | 16+4g..   (head B: base already shifted by -8 dims)
      k1[t] = ld_half<T>(row + 64);                            // dims 32+4g.. | none
    }
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      f32x4 acc = (t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f};
      acc = Tr<T>::mfma(k0[t], q[0], acc);
      acc = Tr<T>::mfma(k1[t], q[1], acc);
      st[0][t] = acc;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // V^T fragments: 3 key steps x 3 head-dim tiles; the last step has keys 64..79 only
  V8 va[NPS][3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const char* row = vb + u * 16 * VS;
    va[0][u] = ld_pair<T>(row, row + 32);
    va[1][u] = ld_pair<T>(row + 64, row + 96);
    va[2][u] = ld_half<T>(row + 128);
  }
  filler();
  softmax_biased(st[0], sl2e, false);                          // denominator comes out of the ones row of V^T
  V8 pb[NPS];
  tiles_to_b<T>(st[0], pb);
  f32x4 o[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NPS; ++s) acc = Tr<T>::mfma(va[s][u], pb[s], acc);
    o[u] = acc;
  }
  // row D = 40 of O^T = tile 2, row 8 = lane row g = 2, register 0
  const float inv = __builtin_amdgcn_rcpf(__shfl(o[2][0], 32 + (int)(threadIdx.x & 15)));
  const float wi = w * inv;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    if (KIND == 0) au[u] = o[u] * inv;
    else if (KIND == 1) ac[u] = o[u] * inv;
    else ac[u] = o[u] * wi + (ac[u] - au[u] * w);
  }
}

// Workgroup = NWV waves x 16 pixels, one HEAD PAIR, one image; walks `iters` strided pixel tiles.
template <typename T, int NWV, int RING>
__global__ __launch_bounds__(64 * NWV, NWV == 12 ? 3 : 2) void xattn_fwd_proj_pair_kernel(const P2 p) {
  using V8 = typename Tr<T>::V8;
  constexpr int TP = 16 * NWV;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int PAIRS = p.H >> 1;
  // block -> (image, tile group, pair): XCD-contiguous over the grid; the PAIRS workgroups of a tile group share y rows
  const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
  const int Lg = xcd_remap(lin, (int)(gridDim.x * gridDim.y));
  const int img = Lg / (int)gridDim.x;
  const int L = Lg - img * (int)gridDim.x;
  const int wt = L / PAIRS, pr = L - wt * PAIRS;
  const int N = p.N, C = p.C, K = p.K, nkc = p.nkc, W = p.W;
  const unsigned row_bytes = (unsigned)C * (unsigned)sizeof(T);
  const size_t act = (size_t)2 * N * row_bytes;
  const char* yb = (const char*)p.y + img * act;
  T* ob = (T*)((char*)p.out + img * act);
  const uint8_t* mask = p.mask + (size_t)img * N;
  const float coef_lane = p.coef[(size_t)img * K + min(lane, K > 0 ? K - 1 : 0)];
  const int nwq = NT * nkc;                       // Wq fragments of the pair (1 KiB each)
  char* lds_kv = smem + (size_t)nwq * FRAG;       // [ctx][hp][BLK]

  // ---- prologue: LDS-DMA of the pair's Wq and of every context (both heads: 2 BLK contiguous per context) -----
  stage_frags(p.wq + (size_t)pr * nwq * FRAG, smem, nwq, wv, NWV, lane);
  {
    const size_t ctx_stride = (size_t)PAIRS * 2 * BLK;
    const char* src = p.kv + (size_t)img * (K + 2) * ctx_stride + (size_t)pr * 2 * BLK;
    for (int c = 0; c < K + 2; ++c) stage_frags(src + c * ctx_stride, lds_kv + (size_t)c * 2 * BLK, 2 * BLK / FRAG, wv, NWV, lane);
    // the V^T fragment reads of the last head-dim tile run 1.2 KiB past the last block (rows 41..47, never used):
    // keep those bytes finite
    char* tail = lds_kv + (size_t)(K + 2) * 2 * BLK;
    if ((int)threadIdx.x * 16 < SLACK) *(u32x4*)(tail + threadIdx.x * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  const int mine = (p.tiles - wt + W - 1) / W;
  const int iters = mine < p.iters ? mine : p.iters;
  const __amdgpu_buffer_rsrc_t y_srd = make_srd(yb, (unsigned)act);
  const unsigned row1 = (unsigned)N * row_bytes;
  auto tile_of = [&](int it) -> int { return wt + it * W; };
  auto voff_of = [&](int it) -> unsigned {
    const int px = tile_of(it) * TP + wv * 16 + c16;
    return (it < iters && px < N) ? (unsigned)px * row_bytes + (unsigned)g * 16u : 0xfffffff0u;
  };
  auto mask_of = [&](int it) -> unsigned {
    const int px = tile_of(it) * TP + wv * 16 + c16;
    return mask[(it < iters && px < N) ? px : 0];
  };
  V8 yr0[RING], yr1[RING];
  unsigned voff = voff_of(0), voffn = voff_of(1);
  unsigned mb = mask_of(0);
#pragma unroll
  for (int j = 0; j < RING; ++j) {
    yr0[j] = srd_load16<V8>(y_srd, voff, 64u * j);
    yr1[j] = srd_load16<V8>(y_srd, voff, row1 + 64u * j);
  }
  const f32x4 kb4 = last_tile_bias(g, p.M);
  const float sl2e = p.sl2e;
  const unsigned kmask = (1u << K) - 1u;
  // per-lane offsets into a (ctx, head) block: K row of key c (+ 8g bytes = dims 4g..), V^T row of dim c (+ keys 4g..)
  const int koff = c16 * 2 * D + 8 * g;
  const int vofs = KBYTES + c16 * VS + 8 * g;
  STA_T_INIT();
  STA_T(0);
  wait_dma_and_sync();
  STA_T(1);

  for (int it = 0; it < iters; ++it) {
    if (it == 1) STA_T(2);
    // ---- projection: 5 column tiles x both batch rows ---------------------------------------------------------
    f32x4 qa0[NT], qa1[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      qa0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      qa1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const V8* wf = (const V8*)smem + lane;
    for (int s0 = 0; s0 < nkc; s0 += RING) {
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        const int s = s0 + j;
        V8 a[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) a[u] = wf[(s * NT + u) * 64];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          qa0[u] = Tr<T>::mfma(a[u], yr0[j], qa0[u]);
          qa1[u] = Tr<T>::mfma(a[u], yr1[j], qa1[u]);
        }
        const bool wrap = s + RING >= nkc;
        const unsigned vo = wrap ? voffn : voff;
        const unsigned so = 64u * (unsigned)(wrap ? s + RING - nkc : s + RING);
        yr0[j] = srd_load16<V8>(y_srd, vo, so);
        yr1[j] = srd_load16<V8>(y_srd, vo, row1 + so);
      }
    }
    if (it == 1) STA_T(3);
    const int px_own = tile_of(it) * TP + wv * 16 + c16;
    const bool valid = px_own < N;
    voff = voffn;
    voffn = voff_of(it + 2);
    const unsigned mbn = mask_of(it + 1);
    const unsigned mbits = valid ? (mb & kmask) : 0u;

    // accumulators -> S^T B operands of both heads (rounded to T once). Tile 2 is shared: lane rows g < 2 hold
    // head A's dims 32..39, lane rows g >= 2 head B's dims 0..7; the other head's slots are zeroed HERE.
    const bool lowg = g < 2;
    V8 qA0[2], qA1[2], qB0[2], qB1[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = j & 3;
      // head A: step 0 = tiles 0 | 1, step 1 = tile 2 (g < 2) | zero
      qA0[0][j] = (T)(j < 4 ? qa0[0][r] : qa0[1][r]);
      qA1[0][j] = (T)(j < 4 ? qa1[0][r] : qa1[1][r]);
      qA0[1][j] = j < 4 ? (lowg ? (T)qa0[2][r] : (T)0.0f) : (T)0.0f;
      qA1[1][j] = j < 4 ? (lowg ? (T)qa1[2][r] : (T)0.0f) : (T)0.0f;
      // head B: step 0 = tile 2 (g >= 2) | tile 3, step 1 = tile 4 | zero
      qB0[0][j] = j < 4 ? (lowg ? (T)0.0f : (T)qa0[2][r]) : (T)qa0[3][r];
      qB1[0][j] = j < 4 ? (lowg ? (T)0.0f : (T)qa1[2][r]) : (T)qa1[3][r];
      qB0[1][j] = j < 4 ? (T)qa0[4][r] : (T)0.0f;
      qB1[1][j] = j < 4 ? (T)qa1[4][r] : (T)0.0f;
    }

    // ---- attention + blend, head A then head B --------------------------------------------------------------
    auto head = [&](auto hb_tag, const V8 (&q0)[2], const V8 (&q1)[2]) {
      constexpr int HB = decltype(hb_tag)::value;
      f32x4 au[3], ac[3];
      // head B's K reads start 8 dims (16 bytes) before the row: its slot (step 0, first half) is dims 4g - 8
      const char* blk = lds_kv + HB * BLK;
      const char* kb = blk + koff - (HB ? 16 : 0);
      const char* vb = blk + vofs;
      attend_compact<T, 0, HB>(kb, vb, q0, kb4, sl2e, 0.f, au, ac);
      attend_compact<T, 1, HB>(kb + 2 * BLK, vb + 2 * BLK, q1, kb4, sl2e, 0.f, au, ac);
      for (int i = 0; i < K; ++i) {
        if (!__ballot((mbits >> i) & 1u)) continue;
        const float cw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), i));
        const float w = ((mbits >> i) & 1u) ? cw : 0.f;
        attend_compact<T, 2, HB>(kb + (size_t)(2 + i) * 2 * BLK, vb + (size_t)(2 + i) * 2 * BLK, q1, kb4, sl2e, w, au, ac);
      }
      if (valid) {
        T* obase = ob + (size_t)px_own * C + (2 * pr + HB) * D;
        store_row16<T, 3>(obase, au, g, D);
        store_row16<T, 3>(obase + (size_t)N * C, ac, g, D);
      }
    };
    head(std::integral_constant<int, 0>{}, qA0, qA1);
    if (it == 1) STA_T(4);
    head(std::integral_constant<int, 1>{}, qB0, qB1);
    if (it == 1) STA_T(5);
    mb = mbn;
  }
  STA_T(8);
  STA_T_END();
}

// Software-pipelined variant: the projection of tile it+1 runs INSIDE the attention of tile it — its 10 k-steps are
// handed, 2-3 at a time, to the four mandatory attention sections of a tile (head A / B x contexts 0 / 1) as `filler`
// (see attend_compact), so their MFMAs sit in the same scheduling region as a softmax. Straight-line code: the k-step
// count is a template parameter (NKC = 10 at C = 320, 5 at C = 160), the y ring is RING = 5 deep as before. The
// projection of the tile after the last is computed on zero rows and discarded.
template <typename T, int NWV, int NKC>
__global__ __launch_bounds__(64 * NWV, 2) void xattn_fwd_proj_pair_sp_kernel(const P2 p) {
  using V8 = typename Tr<T>::V8;
  constexpr int RING = 5;
  constexpr int TP = 16 * NWV;
  static_assert(NKC % RING == 0, "k-steps per tile must be a multiple of the ring depth");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int PAIRS = p.H >> 1;
  const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
  const int Lg = xcd_remap(lin, (int)(gridDim.x * gridDim.y));
  const int img = Lg / (int)gridDim.x;
  const int L = Lg - img * (int)gridDim.x;
  const int wt = L / PAIRS, pr = L - wt * PAIRS;
  const int N = p.N, C = p.C, K = p.K, W = p.W;
  const unsigned row_bytes = (unsigned)C * (unsigned)sizeof(T);
  const size_t act = (size_t)2 * N * row_bytes;
  const char* yb = (const char*)p.y + img * act;
  T* ob = (T*)((char*)p.out + img * act);
  const uint8_t* mask = p.mask + (size_t)img * N;
  const float coef_lane = p.coef[(size_t)img * K + min(lane, K > 0 ? K - 1 : 0)];
  constexpr int nwq = NT * NKC;
  char* lds_kv = smem + (size_t)nwq * FRAG;

  stage_frags(p.wq + (size_t)pr * nwq * FRAG, smem, nwq, wv, NWV, lane);
  {
    const size_t ctx_stride = (size_t)PAIRS * 2 * BLK;
    const char* src = p.kv + (size_t)img * (K + 2) * ctx_stride + (size_t)pr * 2 * BLK;
    for (int c = 0; c < K + 2; ++c) stage_frags(src + c * ctx_stride, lds_kv + (size_t)c * 2 * BLK, 2 * BLK / FRAG, wv, NWV, lane);
    char* tail = lds_kv + (size_t)(K + 2) * 2 * BLK;
    if ((int)threadIdx.x * 16 < SLACK) *(u32x4*)(tail + threadIdx.x * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  const int mine = (p.tiles - wt + W - 1) / W;
  const int iters = mine < p.iters ? mine : p.iters;
  const __amdgpu_buffer_rsrc_t y_srd = make_srd(yb, (unsigned)act);
  const unsigned row1 = (unsigned)N * row_bytes;
  auto tile_of = [&](int it) -> int { return wt + it * W; };
  auto voff_of = [&](int it) -> unsigned {
    const int px = tile_of(it) * TP + wv * 16 + c16;
    return (it < iters && px < N) ? (unsigned)px * row_bytes + (unsigned)g * 16u : 0xfffffff0u;
  };
  auto mask_of = [&](int it) -> unsigned {
    const int px = tile_of(it) * TP + wv * 16 + c16;
    return mask[(it < iters && px < N) ? px : 0];
  };
  // ring state: `voff` = rows of the tile being projected, `voffn` = rows of the tile after it
  V8 yr0[RING], yr1[RING];
  unsigned voff = voff_of(0), voffn = voff_of(1);
  unsigned mb = mask_of(0);
#pragma unroll
  for (int j = 0; j < RING; ++j) {
    yr0[j] = srd_load16<V8>(y_srd, voff, 64u * j);
    yr1[j] = srd_load16<V8>(y_srd, voff, row1 + 64u * j);
  }
  const f32x4 kb4 = last_tile_bias(g, p.M);
  const float sl2e = p.sl2e;
  const unsigned kmask = (1u << K) - 1u;
  const int koff = c16 * 2 * D + 8 * g;
  const int vofs = KBYTES + c16 * VS + 8 * g;
  const V8* wf = (const V8*)smem + lane;
  f32x4 qn0[NT], qn1[NT];                          // projection accumulators of the tile being projected
  // k-steps [S0, S1) of that projection (compile-time range): A fragments from LDS, 10 MFMAs and the ring refill each
  auto ksteps = [&](auto s0_tag, auto s1_tag) {
    constexpr int S0 = decltype(s0_tag)::value, S1 = decltype(s1_tag)::value;
#pragma unroll
    for (int s = S0; s < S1; ++s) {
      const int j = s % RING;
      V8 a[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) a[u] = wf[(s * NT + u) * 64];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        qn0[u] = Tr<T>::mfma(a[u], yr0[j], qn0[u]);
        qn1[u] = Tr<T>::mfma(a[u], yr1[j], qn1[u]);
      }
      const bool wrap = s + RING >= NKC;           // compile time
      const unsigned vo = wrap ? voffn : voff;
      const unsigned so = 64u * (unsigned)(wrap ? s + RING - NKC : s + RING);
      yr0[j] = srd_load16<V8>(y_srd, vo, so);
      yr1[j] = srd_load16<V8>(y_srd, vo, row1 + so);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  // split of the NKC k-steps over the four mandatory attention sections
  using I1 = std::integral_constant<int, (NKC * 1) / 4>;
  using I2 = std::integral_constant<int, (NKC * 2) / 4>;
  using I3 = std::integral_constant<int, (NKC * 3) / 4>;
  using I4 = std::integral_constant<int, NKC>;
  auto zero_q = [&]() {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      qn0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      qn1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  wait_dma_and_sync();
  zero_q();
  ksteps(I0{}, I4{});                              // tile 0, not overlapped with anything
  voff = voffn;
  voffn = voff_of(2);

  for (int it = 0; it < iters; ++it) {
    const int px_own = tile_of(it) * TP + wv * 16 + c16;
    const bool valid = px_own < N;
    const unsigned mbn = mask_of(it + 1);
    const unsigned mbits = valid ? (mb & kmask) : 0u;
    // this tile's q operands from the finished accumulators; the accumulators then start the next tile
    const bool lowg = g < 2;
    V8 qA0[2], qA1[2], qB0[2], qB1[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = j & 3;
      qA0[0][j] = (T)(j < 4 ? qn0[0][r] : qn0[1][r]);
      qA1[0][j] = (T)(j < 4 ? qn1[0][r] : qn1[1][r]);
      qA0[1][j] = j < 4 ? (lowg ? (T)qn0[2][r] : (T)0.0f) : (T)0.0f;
      qA1[1][j] = j < 4 ? (lowg ? (T)qn1[2][r] : (T)0.0f) : (T)0.0f;
      qB0[0][j] = j < 4 ? (lowg ? (T)0.0f : (T)qn0[2][r]) : (T)qn0[3][r];
      qB1[0][j] = j < 4 ? (lowg ? (T)0.0f : (T)qn1[2][r]) : (T)qn1[3][r];
      qB0[1][j] = j < 4 ? (T)qn0[4][r] : (T)0.0f;
      qB1[1][j] = j < 4 ? (T)qn1[4][r] : (T)0.0f;
    }
    zero_q();

    auto head = [&](auto hb_tag, const V8 (&q0)[2], const V8 (&q1)[2], auto f0, auto f1) {
      constexpr int HB = decltype(hb_tag)::value;
      f32x4 au[3], ac[3];
      const char* blk = lds_kv + HB * BLK;
      const char* kb = blk + koff - (HB ? 16 : 0);
      const char* vb = blk + vofs;
      attend_compact<T, 0, HB>(kb, vb, q0, kb4, sl2e, 0.f, au, ac, f0);
      attend_compact<T, 1, HB>(kb + 2 * BLK, vb + 2 * BLK, q1, kb4, sl2e, 0.f, au, ac, f1);
      for (int i = 0; i < K; ++i) {
        if (!__ballot((mbits >> i) & 1u)) continue;
        const float cw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), i));
        const float w = ((mbits >> i) & 1u) ? cw : 0.f;
        attend_compact<T, 2, HB>(kb + (size_t)(2 + i) * 2 * BLK, vb + (size_t)(2 + i) * 2 * BLK, q1, kb4, sl2e, w, au, ac);
      }
      if (valid) {
        T* obase = ob + (size_t)px_own * C + (2 * pr + HB) * D;
        store_row16<T, 3>(obase, au, g, D);
        store_row16<T, 3>(obase + (size_t)N * C, ac, g, D);
      }
    };
    head(std::integral_constant<int, 0>{}, qA0, qA1, [&]() { ksteps(I0{}, I1{}); }, [&]() { ksteps(I1{}, I2{}); });
    head(std::integral_constant<int, 1>{}, qB0, qB1, [&]() { ksteps(I2{}, I3{}); }, [&]() { ksteps(I3{}, I4{}); });
    voff = voffn;
    voffn = voff_of(it + 3);
    mb = mbn;
  }
}

template <typename T, int NWV, int RING>
int launch_pair_cfg(P2 p, int n_img, hipStream_t st) {
  constexpr int TP = 16 * NWV;
  const int pairs = p.H / 2;
  p.tiles = (p.N + TP - 1) / TP;
  long wg = 256L / ((long)pairs * n_img);          // workgroups per (pair, image): one round of one workgroup per CU
  if (wg < 1) wg = 1;
  if (wg > p.tiles) wg = p.tiles;
  p.iters = (int)((p.tiles + wg - 1) / wg);
  if (const int v = g_sta_opt[STA_OPT_STAGED_TILES]) p.iters = v < p.tiles ? v : p.tiles;
  p.W = (p.tiles + p.iters - 1) / p.iters;
  const int lds = lds_bytes(p.C, p.K);
  static StaLdsAttr attr;
  if (!attr.ensure((const void*)xattn_fwd_proj_pair_kernel<T, NWV, RING>, 160 * 1024))
    return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(fwd proj pair) failed");
  hipLaunchKernelGGL((xattn_fwd_proj_pair_kernel<T, NWV, RING>), dim3(p.W * pairs, n_img), dim3(64 * NWV), lds, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "fwd proj pair launch: %s", hipGetErrorString(e));
}

template <typename T, int NWV, int NKC>
int launch_pair_sp(P2 p, int n_img, hipStream_t st) {
  constexpr int TP = 16 * NWV;
  const int pairs = p.H / 2;
  p.tiles = (p.N + TP - 1) / TP;
  long wg = 256L / ((long)pairs * n_img);
  if (wg < 1) wg = 1;
  if (wg > p.tiles) wg = p.tiles;
  p.iters = (int)((p.tiles + wg - 1) / wg);
  if (const int v = g_sta_opt[STA_OPT_STAGED_TILES]) p.iters = v < p.tiles ? v : p.tiles;
  p.W = (p.tiles + p.iters - 1) / p.iters;
  const int lds = lds_bytes(p.C, p.K);
  static StaLdsAttr attr;
  if (!attr.ensure((const void*)xattn_fwd_proj_pair_sp_kernel<T, NWV, NKC>, 160 * 1024))
    return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(fwd proj pair sp) failed");
  hipLaunchKernelGGL((xattn_fwd_proj_pair_sp_kernel<T, NWV, NKC>), dim3(p.W * pairs, n_img), dim3(64 * NWV), lds, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "fwd proj pair sp launch: %s", hipGetErrorString(e));
}

template <typename T, int NWV>
int launch_pair(const P2& p, int n_img, hipStream_t st) {
  if constexpr (NWV == 8) {      // software-pipelined build (projection of the next tile inside this tile's attention)
    if (g_sta_opt[STA_OPT_PROJ_RING] == 2) {
      if (p.nkc == 10) return launch_pair_sp<T, 8, 10>(p, n_img, st);
      if (p.nkc == 5) return launch_pair_sp<T, 8, 5>(p, n_img, st);
    }
  }
  // ring depth 5 ships; 10 (the whole y row of the next tile in flight, 202 registers) measured slower here too
  // (77.6 - 81.3 vs 73.0 - 76.3 us): the projection phase is not waiting for y any more (profiles/r02_proj_fusion.md)
  if constexpr (NWV != 12) {
    if (p.nkc % 10 == 0 && g_sta_opt[STA_OPT_PROJ_RING] == 10) return launch_pair_cfg<T, NWV, 10>(p, n_img, st);
  }
  return launch_pair_cfg<T, NWV, 5>(p, n_img, st);
}

}  // namespace

#ifdef STA_TRACE
extern "C" int sta_debug_set_trace_pair(void* buf) {      // trace build only (tools/trace_proj.py)
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

namespace sta_pair {

int pack_kv(const void* k, const void* v, void* packed, int n_ctx, int M, int C, int heads, int dtype, hipStream_t st) {
  const dim3 grid(n_ctx * heads);
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(pack_kv_pair_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)k, (const __bf16*)v, (__bf16*)packed, M, C, heads);
  else
    hipLaunchKernelGGL(pack_kv_pair_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16*)k, (const _Float16*)v, (_Float16*)packed, M, C, heads);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "pack_kv_pair launch: %s", hipGetErrorString(e));
}

int forward(const void* y, const void* wq_pair, const void* kv_pair, const uint8_t* mask, const float* coef, void* out, int n_img,
            int N, int C, int heads, int M, int K, float sl2e, int dtype, hipStream_t st) {
  P2 p{};
  p.y = y; p.wq = (const char*)wq_pair; p.kv = (const char*)kv_pair; p.mask = mask; p.coef = coef; p.out = out;
  p.N = N; p.C = C; p.H = heads; p.M = M; p.K = K; p.nkc = C / 32; p.sl2e = sl2e;
  const int nwv = g_sta_opt[STA_OPT_STAGED_WAVES] == 4 ? 4 : (g_sta_opt[STA_OPT_STAGED_WAVES] == 12 ? 12 : 8);
  if (dtype == STA_BF16)
    return nwv == 4 ? launch_pair<__bf16, 4>(p, n_img, st) : (nwv == 12 ? launch_pair<__bf16, 12>(p, n_img, st) : launch_pair<__bf16, 8>(p, n_img, st));
  return nwv == 4 ? launch_pair<_Float16, 4>(p, n_img, st) : (nwv == 12 ? launch_pair<_Float16, 12>(p, n_img, st) : launch_pair<_Float16, 8>(p, n_img, st));
}

}  // namespace sta_pair
