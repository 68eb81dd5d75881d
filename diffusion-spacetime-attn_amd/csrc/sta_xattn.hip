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
#include "sta_internal.h"
#include "sta_xattn_dev.h"

namespace {
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
      // forward V^T only: the first padded head-dim row (d % 16 != 0) is a row of ones over the real keys, so the
      // PV MFMAs of the forward kernels also produce sum_key P — the softmax denominator — as row d of O^T
      else if (!second && key < M && dd == d) x = (T)1.0f;
    }
    val[j] = x;
  }
  typename Tr<T>::V8* dst =
      (typename Tr<T>::V8*)((char*)packed + ((size_t)ch * (2 * nfwd) + frag) * FRAG) + lane;
  *dst = val;
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
//
// The kernel is LATENCY bound, not bandwidth bound (in-kernel s_memtime timeline, tools/trace_fwd.py:
// a first-touch global round trip costs 1000-1800 cycles on a freshly launched workgroup while all
// MFMAs of a wave take ~500). So the structure is: every global load the wave will ever need is
// requested in the prologue, oldest first — disc-mask bytes, blend weights for the epilogue, Q, then
// the K-side and V-side fragments — and nothing afterwards waits for memory again:
//   prologue loads -> [wait: mask] tile test -> [wait: Q,K] S^T MFMAs -> softmax -> [wait: V] O^T MFMAs
//   -> fp32 partial to LDS -> one barrier -> combine (LDS only) -> 16-byte stores.
// Local waves request their fragments speculatively (before the tile test) — a wasted 10-25 KB of L2
// reads when the tile misses the disc, in exchange for one round trip less when it does not.
constexpr int NSLOT = 5;   // LDS slots: 0 = A_u (row 0), 1 + w = row-1 partial of wave w
template <typename T, int NDT, int QT>
__global__ __launch_bounds__(256) void xattn_fwd_kernel(const Params pin) {
  using V8 = typename Tr<T>::V8;
  const Params p = for_image<T, NDT>(pin, blockIdx.y, (size_t)(pin.K + 2) * pin.H * pin.N * pin.M);
  constexpr int NKS = nks_of(NDT);
  constexpr int NKF = NKT * NKS, NVF = NPS * NDT;
  constexpr int CHK = 2 * NDT;                    // 8-channel chunks per pixel (compile-time bound; 8*ch < d is checked)
  constexpr int ITEMS = (16 * QT * CHK + 255) / 256;        // combine items per thread
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  // Two block -> (tile, head) maps, chosen per level on measured HBM traffic (profiles/r01_pmc_traffic.md):
  // tile-contiguous ranges per XCD keep the 8 heads of a pixel tile (which share the 128-B lines of the
  // [N][C] rows) on one L2 but make all 8 L2s fetch the whole K/V image; head-major (block b -> head
  // b % H, i.e. XCD b % 8 owns head b % 8) fetches each head's image once and re-fetches partial q lines.
  const int L = p.head_major ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  int tile, h;
  if (p.H == 8) { tile = L >> 3; h = L & 7; } else { tile = L / p.H; h = L % p.H; }   // no integer divide on the hot map
  const int N = p.N, C = p.C, d = p.d, K = p.K;
  const int px0 = tile * 16 * QT;
  const int DP = d + 4;                           // fp32 row stride of a slot: conflict-free b128 writes
  const int slot_floats = 16 * QT * DP;
  float* slots = (float*)smem;
  STA_T_INIT();
  STA_T(0);

  // ---- prologue: request everything, oldest first ------------------------------------------------
  // Every load is UNCONDITIONAL (clamped or bounds-checked address) and the predicate is applied to the
  // value afterwards: a guarded load (`cond ? *p : 0`) makes hipcc emit a branch plus an s_waitcnt per
  // load, i.e. one serialized round trip each. With K == 0 the host points mask/coef at q (readable,
  // ignored). Loads go through buffer descriptors: one instruction each, no per-load 64-bit VALU math.
  const unsigned row_bytes = (unsigned)C * (unsigned)sizeof(T);
  const __amdgpu_buffer_rsrc_t q_srd = make_srd(p.q, 2u * (unsigned)N * row_bytes);
  const __amdgpu_buffer_rsrc_t kv_srd = make_srd(p.packed + (size_t)h * all_frags(NDT) * FRAG,
                                                 (unsigned)(K + 2) * (unsigned)p.H * all_frags(NDT) * FRAG);
  const unsigned ctx_stride = (unsigned)p.H * all_frags(NDT) * FRAG;

  // (a) blend weights: ONE vector load (lane i holds coef[i]) — a scalar load per object would be K
  //     serialized scalar-cache round trips; (b) mask bits of the tile pixels (lane <-> pixel) and of
  //     this thread's epilogue items
  float coefv[MAXK];
  const float coef_lane = p.coef[min(lane, K > 0 ? K - 1 : 0)];
  unsigned mbits = p.mask[min(px0 + lane, N - 1)];
  unsigned mi[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) mi[j] = p.mask[min(px0 + (int)(threadIdx.x + 256 * j) / CHK, N - 1)];

  // (c) operands of this wave's first context (speculative for local contexts). Q: B operand, 16 B per
  //     lane at d-offset 32s + 8g of its pixel row; K/V: fragment f at byte f*1024 + lane*16 of the image
  V8 qf[QT][NKS], ka[NKF], va[NVF];
  int c = wv;
  auto request = [&](int cc) {
    const unsigned qrow = cc == 0 ? 0u : (unsigned)N * row_bytes;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int px = px0 + 16 * qt + c16;
      // pixels >= N and head-dim offsets >= d are pushed out of the descriptor's range -> read as 0
      const unsigned base = px < N ? (unsigned)px * row_bytes + (unsigned)(h * d + 8 * g) * (unsigned)sizeof(T) : 0xfffffff0u;
#pragma unroll
      for (int s = 0; s < NKS; ++s)
        qf[qt][s] = srd_load16<V8>(q_srd, (32 * s + 8 * g < d) ? base : 0xfffffff0u, qrow + 64u * s);
    }
    const unsigned cbase = (unsigned)cc * ctx_stride;
#pragma unroll
    for (int f = 0; f < NKF; ++f) ka[f] = srd_load16<V8>(kv_srd, lane * 16, cbase + f * FRAG);
#pragma unroll
    for (int f = 0; f < NVF; ++f) va[f] = srd_load16<V8>(kv_srd, lane * 16, cbase + (NKF + f) * FRAG);
  };
  if (c < K + 2) request(c);
  __builtin_amdgcn_sched_barrier(0);
  STA_T(1);

  // first consumers of the (oldest) mask loads: which discs touch this tile — every wave computes the
  // same answer, so no LDS flag and no barrier — and the epilogue's per-item weight sums
  unsigned tile_bits = 0;
  float wsum[ITEMS];
  {
    const bool in_tile = lane < 16 * QT && px0 + lane < N;
    mbits = in_tile ? (mbits & ((1u << K) - 1u)) : 0u;
#pragma unroll
    for (int i = 0; i < MAXK; ++i) {
      coefv[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), i));
      if (i < K && __any((mbits >> i) & 1u)) tile_bits |= 1u << i;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      wsum[j] = 0.f;
#pragma unroll
      for (int i = 0; i < MAXK; ++i) wsum[j] += (i < K && ((mi[j] >> i) & 1u)) ? coefv[i] : 0.f;
    }
  }
  if (p.aux) tile_bits = (1u << K) - 1u;          // parity mode: every map is wanted
  // predicate-free softmax (see softmax_biased) unless maps are wanted or whole key tiles are padding
  const bool fast = !p.aux && p.M > 16 * (NKT - 1);
  const f32x4 kb4 = fast ? last_tile_bias(g, p.M) : f32x4{0.f, 0.f, 0.f, 0.f};
  STA_T(2);

  f32x4 part[QT][NDT];                            // this wave's row-1 partial (wave 0, ctx 0: A_u)
  bool have_part = false;
  while (c < K + 2) {
    const bool active = c < 2 || ((tile_bits >> (c - 2)) & 1u);
    if (active) {
      // S^T = K Q^T : every A fragment is used for QT pixel tiles
      f32x4 st[QT][NKT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < NKT; ++t) st[qt][t] = (t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int s = 0; s < NKS; ++s)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) st[qt][t] = Tr<T>::mfma(ka[t * NKS + s], qf[qt][s], st[qt][t]);
      STA_T(3);
      STA_T(4);

      float wc[QT];
      V8 pb[QT][NPS];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const float inv = fast ? softmax_biased(st[qt], p.sl2e) : softmax_keys_fast(st[qt], g, p.M, p.sl2e);
        if (p.aux && px0 + 16 * qt + c16 < N) {
          float* mrow = p.aux + (((size_t)c * p.H + h) * N + (px0 + 16 * qt + c16)) * p.M;
#pragma unroll
          for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = 16 * t + 4 * g + r;
              if (key < p.M) mrow[key] = st[qt][t][r] * inv;
            }
        }
        tiles_to_b<T>(st[qt], pb[qt]);
        wc[qt] = inv;
        if (c >= 2) {   // this context's blend weight for the lane's pixel: coef_i * mask_i(px)
          const unsigned m = ((unsigned)__shfl((int)mbits, 16 * qt + c16) >> (c - 2)) & 1u;   // 0 for pixels >= N
          const float cw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), c - 2));
          wc[qt] = m ? inv * cw : 0.f;
        }
      }
      STA_T(5);

      // O^T = V^T P^T, then this context's share of the blend
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
    c += 4;
    if (c < K + 2) request(c);                    // K > 2 only: next context of this wave
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
  for (int cc = 1; cc < K + 2; ++cc)
    if (cc < 2 || ((tile_bits >> (cc - 2)) & 1u)) slot_bits |= 1u << (cc & 3);
  __syncthreads();
  STA_T(7);

  // combine + store: thread -> (pixel, 8-channel chunk); LDS reads only, 16-byte stores to both rows
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int it = threadIdx.x + 256 * j;
    const int pl = it / CHK, ch = it - pl * CHK;
    const int px = px0 + pl;
    if (pl >= 16 * QT || 8 * ch >= d || px >= N) continue;
    const float* s0 = slots + pl * DP + 8 * ch;
    float au[8], o1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      au[e] = s0[e];
      o1[e] = -wsum[j] * au[e];
    }
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if ((slot_bits >> w) & 1u) {
        const float* sw = slots + (1 + w) * slot_floats + pl * DP + 8 * ch;
#pragma unroll
        for (int e = 0; e < 8; ++e) o1[e] += sw[e];
      }
    V8 r0, r1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      r0[e] = (T)au[e];
      r1[e] = (T)o1[e];
    }
    T* ob = (T*)p.out + (size_t)px * C + h * d + 8 * ch;
    *(V8*)ob = r0;
    *(V8*)(ob + (size_t)N * C) = r1;
  }
  STA_T(8);
  STA_T_END();
}

// --------------------------------------------------------------------------------------------------
// forward, LDS-resident variant for the large levels (many pixel tiles per head, d <= 96)
// --------------------------------------------------------------------------------------------------
// With thousands of pixel tiles per head the per-CU vector-memory path (64 B/clk) becomes the limit
// of the wave-per-context kernel above: every wave streams its context's fragments itself. Here a
// workgroup (4 waves x 16 pixels, one head) copies the fragments of ALL contexts it needs into LDS
// once (LDS-DMA, 76 KB at d = 40 / K = 2), passes ONE barrier, and each wave then attends its 16
// pixels against every context from LDS (2x the bandwidth of the vector-memory path, no L2 latency).
// LDS fragment reads are hoisted into registers ahead of the MFMAs for the same reason the global
// loads are in the other kernel. Contexts 0/1 are requested before the disc mask is known; local
// contexts right after the tile test. If the contexts do not fit LDS at once they go in groups.
// A workgroup keeps the fragments of one head in LDS and walks `p.iters` consecutive pixel tiles of
// 16*NWV*QT pixels with them: the staging traffic (L2 -> LDS, 19-55 KB per context) and the staging
// latency are paid once per workgroup instead of once per 64/128 pixels — at 8 images per launch the
// re-staging was MORE bytes through the per-CU vector-memory path than q and out together. After the
// one barrier the waves run independently (no barrier per tile): while one wave waits for the q rows of
// its next tile, the other waves of the SIMD compute.
// MAXIT = 1 is the single-tile build for launches with few workgroups (one image): no mask bytes beyond the
// tile's own, no prefetch code.
constexpr int STAGED_MAXIT = 12;
// LL2 (round 6; d > 64 where the K + 2 contexts of a head exceed a CU's LDS: SD-v1 level 2 / mid at K >= 1, level 1 from K = 4):
// only the two mandatory contexts are staged; a local context is read as MFMA operands straight from the packed image in L2
// (SrdFrags: one coalesced 1-KiB buffer load per fragment) by the waves whose pixels a disc touches. No second staging group, no
// barrier behind the first one, and the workgroup keeps its image for `iters` tiles — the grouped variant (LL2 = false) re-staged
// 2 x 55 KiB per 128 pixels at d = 160 behind two barriers per tile.
template <typename T, int NDT, int QT, int NWV, int MAXIT, bool LL2 = false>
__global__ __launch_bounds__(64 * NWV, NWV == 12 ? 3 : 1) void xattn_fwd_staged_kernel(const Params pin) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  // Block -> (image, tile group, head). Workgroups are dispatched x-fastest and land on XCD (x + gridDim.x * y) % 8, so
  // the XCD-contiguous remap runs over the WHOLE grid, images included: with many tiles per workgroup gridDim.x
  // is small (16 at 16 images per launch) and a per-image remap would spread the 8 heads of a tile group — which
  // share the 128-B lines of the q rows — over 4 L2s (HBM-side traffic 1.42x algorithmic; 1.2x with this map).
  const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
  const int Lg = pin.head_major ? lin : xcd_remap(lin, (int)(gridDim.x * gridDim.y));
  const int img = Lg / (int)gridDim.x;
  const Params p = for_image<T, NDT>(pin, img, 0);
  constexpr int NKS = nks_of(NDT);
  constexpr int NKF = NKT * NKS, NVF = NPS * NDT, NFWD = NKF + NVF;
  constexpr int CB = NFWD * FRAG;                 // bytes of one staged context
  constexpr int TP = 16 * NWV * QT;               // pixels per tile: NWV waves x QT sub-tiles x 16
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int L = Lg - img * (int)gridDim.x;
  int wt, h;
  if (p.H == 8) { wt = L >> 3; h = L & 7; } else { wt = L / p.H; h = L % p.H; }
  const int N = p.N, C = p.C, d = p.d, K = p.K;
  // tiles of this workgroup: wt, wt + W, wt + 2W, ... (W = workgroups per head). Strided, not consecutive:
  // the discs are spatially compact, so consecutive tiles would make some workgroups all-local (3-4
  // contexts per pixel) and others all-global (2) — measured lifetimes 18-30 us at 8 images per launch.
  const int W = p.ntiles, tiles = p.tiles;
  const int iters = MAXIT == 1 ? 1 : ((tiles - wt + W - 1) / W < p.iters ? (tiles - wt + W - 1) / W : p.iters);
  const int G = p.ntiles_aux;                     // contexts that fit LDS at once (>= 2)
  STA_T_INIT();
  STA_T(0);

  const size_t ctx_stride = (size_t)p.H * all_frags(NDT) * FRAG;
  const char* img_h = p.packed + (size_t)h * all_frags(NDT) * FRAG;

  // ---- prologue: oldest first — weights, mask bits of every tile, fragments of contexts 0 and 1, Q ----
  const float coef_lane = p.coef[min(lane, K > 0 ? K - 1 : 0)];
  constexpr int CPT = TP / 64;                     // 64-pixel chunks per tile
  constexpr int MAXCH = MAXIT * CPT;
  unsigned span_bits = 0;                          // OR of the mask bytes of this workgroup's pixels
  {
    unsigned tb[MAXCH];
#pragma unroll
    for (int j = 0; j < MAXCH; ++j)                // unconditional, clamped
      tb[j] = p.mask[min((wt + (j / CPT) * W) * TP + 64 * (j % CPT) + lane, N - 1)];
#pragma unroll
    for (int j = 0; j < MAXCH; ++j)
      span_bits |= (j / CPT < iters && (wt + (j / CPT) * W) * TP + 64 * (j % CPT) + lane < N) ? tb[j] : 0u;
  }
  auto stage_first_group = [&]() {
    stage_frags(img_h, smem, NFWD, wv, NWV, lane);
    stage_frags(img_h + ctx_stride, smem + CB, NFWD, wv, NWV, lane);
  };
  stage_first_group();
  // a wave's QT sub-tiles of its it-th tile: pixels (wt + it*W)*TP + (wv*QT + qt)*16 + c16, rows 0 (uncond)
  // and 1 (cond). The q rows (and mask byte) of tile it+1 are requested before tile it is computed.
  V8 q0[QT][NKS], q1[QT][NKS], q1n[QT][NKS];
  unsigned mb[QT], mbn[QT];
  auto request_q0 = [&](int it, V8 (&a0)[QT][NKS]) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int px = (wt + it * W) * TP + (wv * QT + qt) * 16 + c16;
      const bool ok = it < iters && px < N;
      load_b_frags<T, NKS>((const T*)p.q + (size_t)(ok ? px : 0) * C + h * d, ok, g, d, a0[qt]);
    }
  };
  auto request_q1 = [&](int it, V8 (&a1)[QT][NKS], unsigned (&m)[QT]) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int px = (wt + it * W) * TP + (wv * QT + qt) * 16 + c16;
      const bool ok = it < iters && px < N;
      m[qt] = p.mask[ok ? px : 0];
      load_b_frags<T, NKS>((const T*)p.q + ((size_t)N + (ok ? px : 0)) * C + h * d, ok, g, d, a1[qt]);
    }
  };
  request_q0(0, q0);
  request_q1(0, q1, mb);
  __builtin_amdgcn_sched_barrier(0);
  STA_T(1);

  // which discs touch the tiles (workgroup-uniform: every wave looked at the same bytes)
  unsigned tile_bits = 0;
  span_bits &= (1u << K) - 1u;
  for (int i = 0; i < K; ++i)                     // K iterations, not MAXK predicated ones: fewer issue slots
    if (__ballot((span_bits >> i) & 1u)) tile_bits |= 1u << i;
  // active contexts in order: 0, 1, then the local ones whose disc touches a tile; entry e of that list
  // sits in LDS slot e % G. Stage the locals that still fit beside contexts 0 and 1.
  auto stage_locals = [&](unsigned bits, int first_slot, int count) {   // lowest `count` set bits of `bits`
    for (int n = 0; n < count && bits; ++n) {
      const int i = __builtin_ctz(bits);
      bits &= bits - 1;
      stage_frags(img_h + (size_t)(2 + i) * ctx_stride, smem + (first_slot + n) * CB, NFWD, wv, NWV, lane);
    }
  };
  if (!LL2) stage_locals(tile_bits, 2, G - 2);
  const bool resident = LL2 || 2 + __builtin_popcount(tile_bits) <= G;   // everything stays in LDS for all tiles
  SrdFrags<V8> gfr;
  if constexpr (LL2) {
    gfr.r = make_srd(p.packed, (unsigned)((K + 2) * ctx_stride));
    gfr.voff = (unsigned)lane * 16u;
    gfr.soff = 0u;
  }
  const f32x4 kb4 = last_tile_bias(g, p.M);
  const float sl2e = p.sl2e;
  const int sumrow = (d & 15) ? (d & 15) : -1;    // packed V^T carries a ones row at head-dim index d (pack_kv_kernel)
  STA_T(2);
  wait_dma_and_sync();
  STA_T(3);

  for (int it = 0; it < iters; ++it) {
    if (it == 1) STA_T(7);
    if (it > 0 && !resident) {   // the first group was overwritten by a later one: bring it back
      __syncthreads();
      stage_first_group();
      stage_locals(tile_bits, 2, G - 2);
      wait_dma_and_sync();
    }
    if (MAXIT > 1) request_q1(it + 1, q1n, mbn);
    if (it == 1) STA_T(9);
    f32x4 au[QT][NDT], ac[QT][NDT];
    float w[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) w[qt] = 0.f;
    attend_staged<T, NDT, QT, 0>((const V8*)smem + lane, q0, kb4, sl2e, w, au, ac, sumrow);
    if (MAXIT > 1) request_q0(it + 1, q0);        // context 0 was q0's only consumer: next tile's rows go in place
    if (it == 0) STA_T(4);
    if (it == 1) STA_T(10);
    attend_staged<T, NDT, QT, 1>((const V8*)(smem + CB) + lane, q1, kb4, sl2e, w, au, ac, sumrow);
    if (it == 0) STA_T(5);
    if (it == 1) STA_T(11);
    unsigned wave_bits = 0;                        // discs that touch THIS wave's pixels of the tile
    bool valid[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      valid[qt] = (wt + it * W) * TP + (wv * QT + qt) * 16 + c16 < N;
      mb[qt] = valid[qt] ? (mb[qt] & tile_bits) : 0u;
      for (unsigned bits = tile_bits; bits; bits &= bits - 1) {   // only the discs that touch the workgroup's tiles
        const int i = __builtin_ctz(bits);
        if (__ballot((mb[qt] >> i) & 1u)) wave_bits |= 1u << i;
      }
    }
    unsigned rest = LL2 ? 0u : tile_bits;
    if constexpr (LL2) {
      for (unsigned bits = wave_bits; bits; bits &= bits - 1) {   // the discs that touch this wave's pixels, operands from L2
        const int i = __builtin_ctz(bits);
        const float cw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), i));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) w[qt] = ((mb[qt] >> i) & 1u) ? cw : 0.f;
        gfr.soff = (unsigned)((size_t)(2 + i) * ctx_stride + (size_t)h * all_frags(NDT) * FRAG);
        attend_staged<T, NDT, QT, 2>(gfr, q1, kb4, sl2e, w, au, ac, sumrow);
      }
    }
    for (int slot = 2; rest;) {
      if (slot == G) {              // next group: everyone is done reading the previous one
        __syncthreads();
        stage_locals(rest, 0, G);
        wait_dma_and_sync();
        slot = 0;
      }
      const int i = __builtin_ctz(rest);
      rest &= rest - 1;
      const int myslot = slot++;
      if (!((wave_bits >> i) & 1u)) continue;     // none of this wave's pixels inside the disc
      const float cw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), i));
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) w[qt] = ((mb[qt] >> i) & 1u) ? cw : 0.f;
      attend_staged<T, NDT, QT, 2>((const V8*)(smem + myslot * CB) + lane, q1, kb4, sl2e, w, au, ac, sumrow);
    }
    if (it == 0) STA_T(6);
    if (it == 1) STA_T(12);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (valid[qt]) {
        T* obase = (T*)p.out + (size_t)((wt + it * W) * TP + (wv * QT + qt) * 16 + c16) * C + h * d;
        store_row16<T, NDT>(obase, au[qt], g, d);
        store_row16<T, NDT>(obase + (size_t)N * C, ac[qt], g, d);
      }
      if (MAXIT > 1) {
        mb[qt] = mbn[qt];
#pragma unroll
        for (int s2 = 0; s2 < NKS; ++s2) q1[qt][s2] = q1n[qt][s2];
      }
    }
  }
  STA_T(8);
  STA_T_END();
}

// --------------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------------
}  // namespace

// error text shared by every translation unit of the library (sta_internal.h)
thread_local char g_sta_err[256] = "";
// kernel-selection overrides (sta_set_option): 0 = automatic. Written by tests/tools only, read at launch.
StaOpt g_sta_opt[STA_OPT_COUNT];
int sta_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_sta_err, sizeof g_sta_err, fmt, ap);
  va_end(ap);
  return code;
}

namespace {
#define g_err g_sta_err
#define fail sta_fail

// Pixel tiles per wave of the wave-per-context kernel. QT > 1 reuses each fragment for more pixels but
// measured slower at every level (register pressure: N=1024 d=80 7.2 -> 8.7 us; 8 images 39.8 -> 52.8 us),
// so QT = 1 ships and larger values stay reachable through the tuning knob only.
int pick_qt(int N, int heads, int ndt, int n_img) {
  const int cap = ndt <= 3 ? 4 : (ndt <= 6 ? 2 : 1);
  if (const int v = g_sta_opt[STA_OPT_SPLIT_QT]) {
    if (v == 1 || v == 2 || v == 4) return v < cap ? v : cap;
  }
  (void)N; (void)heads; (void)n_img;
  return 1;
}

template <typename T, int NDT, int QT>
int launch_fwd(const Params& p0, hipStream_t st) {
  Params p = p0;
  p.ntiles = (p.N + 16 * QT - 1) / (16 * QT);
  // 7 extra copies of the fragment image (one per further XCD) vs the partial-line over-fetch of q
  // (a head's d*2-byte segment of each row straddles 128-B lines): head-major wins from d = 80 up
  p.head_major = (p.H % 8 == 0 && NDT >= 5) ? 1 : 0;
  if (g_sta_opt[STA_OPT_HEAD_MAJOR]) p.head_major = g_sta_opt[STA_OPT_HEAD_MAJOR] == 1 ? 1 : 0;
  const int lds = NSLOT * 16 * QT * (p.d + 4) * (int)sizeof(float);
  static StaLdsAttr attr;
  constexpr int lds_max = NSLOT * 16 * QT * (16 * NDT + 4) * (int)sizeof(float);
  if (!attr.ensure((const void*)xattn_fwd_kernel<T, NDT, QT>, lds_max)) return fail(STA_E_LAUNCH, "hipFuncSetAttribute(fwd) failed");
  hipLaunchKernelGGL((xattn_fwd_kernel<T, NDT, QT>), dim3(p.ntiles * p.H, p.n_img), dim3(256), lds, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : fail(STA_E_LAUNCH, "fwd launch: %s", hipGetErrorString(e));
}

template <typename T, int NDT, int QT, int NWV>
int launch_fwd_staged_cfg(const Params& p0, hipStream_t st) {
  constexpr int CB = fwd_frags(NDT) * FRAG;
  constexpr int TP = 16 * NWV * QT;
  Params p = p0;
  p.head_major = (p.H % 8 == 0 && NDT >= 5) ? 1 : 0;   // one head per XCD: its K/V image is fetched by one L2 only
  if (g_sta_opt[STA_OPT_HEAD_MAJOR]) p.head_major = g_sta_opt[STA_OPT_HEAD_MAJOR] == 1 ? 1 : 0;
  int G = (150 * 1024) / CB;                     // leave room: 160 KiB LDS per CU
  if (G > p.K + 2) G = p.K + 2;
  // Tiles per workgroup: enough that ONE round of workgroups (LDS-, wave- and register-limited residency on
  // the 256 CUs) covers the launch, at most STAGED_MAXIT; 1 when the contexts do not fit LDS together and are staged in groups.
  const int tiles = (p.N + TP - 1) / TP;
  auto iters_for = [&](int lds_bytes) {
    int per_cu = (160 * 1024) / lds_bytes;
    if (per_cu > 32 / NWV) per_cu = 32 / NWV;
    if (NWV == 12) per_cu = 1;                   // 3 waves per SIMD by register budget
    if (per_cu < 1) per_cu = 1;
    long wg_per_head = (256L * per_cu) / ((long)p.H * p.n_img);   // workgroups per (head, image) in one round
    if (wg_per_head < 1) wg_per_head = 1;
    int it = (int)((tiles + wg_per_head - 1) / wg_per_head);
    return it > STAGED_MAXIT ? STAGED_MAXIT : (it < 1 ? 1 : it);
  };
  // Not every context fits: the two mandatory ones resident, locals from L2 — where a workgroup then walks at least two tiles with its
  // image (level 2 at 64 images: 74.7 -> 68.4 us; level 1 at K = 4, 16 images: -6.5 %). One-tile launches keep the grouped staging: the
  // second group's LDS-DMA beats per-fragment L2 latency there (level 2 at 16 images: 20.1 against 24.1 us in situ).
  // Built for d > 64; STA_OPT_PROJ_LL2 = 2 keeps the grouped staging everywhere (A/B, tests), STA_OPT_STAGED_TILES forces a tile count.
  bool ll2 = NDT >= 5 && G < p.K + 2 && (size_t)(p.K + 2) * p.H * all_frags(NDT) * FRAG < (1ull << 32) && g_sta_opt[STA_OPT_PROJ_LL2] != 2;
  if (ll2 && !g_sta_opt[STA_OPT_STAGED_TILES] && iters_for(2 * CB) < 2) ll2 = false;
  if (ll2) G = 2;
  p.ntiles_aux = G;
  const int lds = G * CB;
  int iters = iters_for(lds);
  if (G < p.K + 2 && !ll2) iters = 1;
  if (const int v = g_sta_opt[STA_OPT_STAGED_TILES]) { if (v >= 1 && v <= STAGED_MAXIT) iters = v; }
  p.iters = iters;
  p.tiles = tiles;
  p.ntiles = (tiles + iters - 1) / iters;
  // multi-tile launches carry several images: with the grid-wide XCD-contiguous map an XCD owns whole (image, tile
  // group) units — all 8 heads, so q lines AND the image's K/V fragments are fetched by one L2 only
  if (iters > 1 && !g_sta_opt[STA_OPT_HEAD_MAJOR]) p.head_major = 0;
  auto launch = [&](auto kernel, StaLdsAttr& attr) {
    if (!attr.ensure((const void*)kernel, 160 * 1024)) return fail(STA_E_LAUNCH, "hipFuncSetAttribute(fwd staged) failed");
    hipLaunchKernelGGL(kernel, dim3(p.ntiles * p.H, p.n_img), dim3(64 * NWV), lds, st, p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? STA_OK : fail(STA_E_LAUNCH, "fwd staged launch: %s", hipGetErrorString(e));
  };
  static StaLdsAttr attr1, attrn;
  if constexpr (NDT >= 5) {
    static StaLdsAttr attr1l, attrnl;
    if (ll2) return iters == 1 ? launch(xattn_fwd_staged_kernel<T, NDT, QT, NWV, 1, true>, attr1l) : launch(xattn_fwd_staged_kernel<T, NDT, QT, NWV, STAGED_MAXIT, true>, attrnl);
  }
  if (iters == 1) return launch(xattn_fwd_staged_kernel<T, NDT, QT, NWV, 1>, attr1);
  return launch(xattn_fwd_staged_kernel<T, NDT, QT, NWV, STAGED_MAXIT>, attrn);
}

// Workgroup shape of the LDS-resident kernel (rocprofv3 durations in profiles/r01_kernel_variants.md).
// 4 waves x 16 px when the launch has few workgroups: latency is what matters. When a launch carries several
// images it is throughput bound and more waves share one LDS image: 12 waves (3 per SIMD, 148 VGPRs: room for
// the q prefetch without spills) at d <= 48, 8 waves above (from d = 112 up with per-tile fragment fetches, which
// is what keeps a wave under 256 registers there). Two tiles per wave (QT = 2) measured slower
// and stays selectable for experiments only.
template <typename T, int NDT>
int launch_fwd_staged(const Params& p, hipStream_t st) {
  const long w64 = (long)((p.N + 63) / 64) * p.H * p.n_img;      // 64-pixel workgroups in the launch
  int qt = 1;
  int nwv = 4;
  if (NDT <= 3 && w64 >= 2048) nwv = 12;
  if (NDT > 3 && w64 >= 512) nwv = 8;             // d = 160: 25.0 -> 18.3 us at 16 images (fragments fetched per tile: 188 VGPRs)
  if (g_sta_opt[STA_OPT_STAGED_QT]) qt = g_sta_opt[STA_OPT_STAGED_QT] == 2 ? 2 : 1;
  if (const int v = g_sta_opt[STA_OPT_STAGED_WAVES]) nwv = v == 8 ? 8 : (v == 12 ? 12 : 4);
  if constexpr (NDT <= 3) {
    if (nwv == 12) return launch_fwd_staged_cfg<T, NDT, 1, 12>(p, st);   // 16 waves at 128 VGPRs spill (88 B/lane): 44 vs 33 us
  }
  if constexpr (NDT <= 6) {
    if (qt == 2) return launch_fwd_staged_cfg<T, NDT, 2, 4>(p, st);
  }
  if (nwv == 8) return launch_fwd_staged_cfg<T, NDT, 1, 8>(p, st);
  return launch_fwd_staged_cfg<T, NDT, 1, 4>(p, st);
}

// Which forward kernel — rocprofv3 kernel durations in us, K = 2, bf16, MI355X (profiles/r01_kernel_variants.md),
// by images per launch I (staged = LDS-resident kernel above with 4 / 8 / 12 waves, split = wave per context):
//   N=4096 d=40 : I=1 staged4 8.4 | split 14-18     I=2 staged4 13.4     I=4 staged12 20.7      I=8 staged12 32.8 | split 93
//   N=1024 d=80 : I=1 split 7.2 | staged4 10.5      I=2 staged4 7.9      I=4 staged8 9.8        I=8 staged8 16.9 | split 40
//   N=256 d=160 : I=1 split 8.0                      I=2 split 8.5        I=4 split 15.0         I=8 staged4 12.5 | split 29
//   N=64  d=160 : split 7.8-8.6 | staged 14-15.7 at every I
// -> staged from 256 64-pixel workgroups per launch (launch_fwd_staged picks the workgroup shape and the tiles per
//    workgroup); the wave-per-context kernel below that, for attention-map output and for M <= 64.
bool use_staged(int N, int heads, int ndt, int n_img) {
  if (g_sta_opt[STA_OPT_FWD_KERNEL] == 1) return true;
  if (g_sta_opt[STA_OPT_FWD_KERNEL] == 2) return false;
  (void)ndt;
  return (long)((N + 63) / 64) * heads * n_img >= 256;
}

template <typename T, int NDT>
int launch_fwd_qt(const Params& p, int qt, hipStream_t st) {
  if constexpr (NDT <= 3) { if (qt == 4) return launch_fwd<T, NDT, 4>(p, st); }
  if constexpr (NDT <= 6) { if (qt >= 2) return launch_fwd<T, NDT, 2>(p, st); }
  return launch_fwd<T, NDT, 1>(p, st);
}

template <typename T>
int dispatch_fwd(const Params& p, hipStream_t st) {
  const int ndt = (p.d + 15) / 16;
  // the LDS-resident kernel has no attention-map output and assumes padded keys in the last key tile only
  if (!p.aux && p.M > 16 * (NKT - 1) && use_staged(p.N, p.H, ndt, p.n_img)) {
    switch (ndt) {
      case 1: return launch_fwd_staged<T, 1>(p, st);
      case 2: return launch_fwd_staged<T, 2>(p, st);
      case 3: return launch_fwd_staged<T, 3>(p, st);
      case 4: return launch_fwd_staged<T, 4>(p, st);
      case 5: return launch_fwd_staged<T, 5>(p, st);
      case 6: return launch_fwd_staged<T, 6>(p, st);
      case 7: return launch_fwd_staged<T, 7>(p, st);
      case 8: return launch_fwd_staged<T, 8>(p, st);
      case 9: return launch_fwd_staged<T, 9>(p, st);
      case 10: return launch_fwd_staged<T, 10>(p, st);
    }
  }
  const int qt = pick_qt(p.N, p.H, ndt, p.n_img);
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

}  // namespace

extern "C" {

#ifdef STA_TRACE
// trace build only: buf[0] = workgroup to trace, buf[8 + wave*16 + i] = s_memtime at point i
int sta_debug_set_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

int sta_version(void) { return STA_VERSION; }

#ifndef STA_BUILT_WITH
#define STA_BUILT_WITH ""
#endif
const char* sta_built_with(void) { return STA_BUILT_WITH; }

int sta_set_option(int key, int value) {
  g_err[0] = 0;
  if (key < 0 || key >= STA_OPT_COUNT) return fail(STA_E_ARG, "unknown option %d", key);
  g_sta_opt[key] = value;
  return STA_OK;
}

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
                  float* maps, int n_img, int N, int C, int heads, int M, int K, float scale, int dtype,
                  void* stream) {
  g_err[0] = 0;
  if (!q || !packed || !out) return fail(STA_E_ARG, "null pointer");
  if (n_img < 1 || n_img > 65535) return fail(STA_E_ARG, "n_img=%d", n_img);
  if (int rc = check_shape(N, C, heads, M, K)) return rc;
  if (K > 0 && (!mask || !coef)) return fail(STA_E_ARG, "mask/coef required when K > 0");
  if (dtype != STA_BF16 && dtype != STA_F16) return fail(STA_E_UNSUP, "dtype %d", dtype);
  Params p{};
  p.q = q; p.packed = (const char*)packed; p.mask = mask; p.coef = coef; p.out = out; p.dout = nullptr;
  if (K == 0) {  // the kernel's prologue loads are unconditional: give it readable (ignored) bytes
    p.mask = (const uint8_t*)q;
    p.coef = (const float*)q;
  }
  p.aux = maps; p.N = N; p.C = C; p.H = heads; p.d = C / heads; p.M = M; p.K = K; p.n_img = n_img;
  p.scale = scale; p.sl2e = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  return dtype == STA_BF16 ? dispatch_fwd<__bf16>(p, st) : dispatch_fwd<_Float16>(p, st);
}


}  // extern "C"
