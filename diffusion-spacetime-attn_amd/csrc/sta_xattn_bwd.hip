// sta_xattn_bwd.hip — backward of the fused spatial-temporal cross-attention (dq, dcoef) for MI355X (gfx950 / CDNA4);
// C-ABI in include/sta_xattn.h (sta_xattn_bwd, sta_xattn_bwd_workspace_bytes). MFMA operand conventions: sta_xattn.hip.
//
// What it replaces (reference file:line, under attention_optimization/stable-diffusion/):
//   ldm/modules/diffusionmodules/util.py:123-145  the part of CheckpointFunction.backward that differentiates
//                                     BasicTransformerBlock._forward's K+1 attn2 calls and their disc-masked,
//                                     coef-weighted blend (ldm/modules/attention.py:278-294) w.r.t. x and coef
//
// Per context with probabilities P (normalised) and upstream gradient G on A = P V:
//   dP = G V^T ; delta = sum_key P dP ; dS = P (dP - delta) ; dQ = scale * dS K          (no dK / dV: the prompts are frozen)
// and for the blend weights, with dP_i = dO1 V_i^T (the UNWEIGHTED upstream of local context i):
//   dcoef_i = sum_px mask_i(px) sum_ch dO1 (A_i - A_u) = sum_px mask_i(px) (sum_key P_i dP_i  -  sum_key P_0 (dO1 V_0^T))
// i.e. the dot products with the attention OUTPUTS are the softmax-weighted sums the dS step needs anyway (delta_i), so the
// LDS-resident kernel below never forms A = P V: three MFMA phases per context (S^T, dP^T, dQ^T) instead of four.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sta_xattn.h"
#include "sta_internal.h"
#include "sta_xattn_dev.h"

namespace {
#define g_err g_sta_err
#define fail sta_fail

// --------------------------------------------------------------------------------------------------
// backward (dq, dcoef)
// --------------------------------------------------------------------------------------------------
// Sum the per-wave partials in a fixed order: one block of 256 threads per object.
__global__ __launch_bounds__(256) void dcoef_reduce_kernel(const float* __restrict__ part, float* dcoef,
                                                           int n) {
  __shared__ float sm[256];
  const float* src = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * n;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += src[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dcoef[blockIdx.y * gridDim.x + blockIdx.x] = sm[0];
}

// --------------------------------------------------------------------------------------------------
// backward (dq, dcoef): ONE LDS-resident multi-tile kernel for every head dim and launch size
// --------------------------------------------------------------------------------------------------
// Rounds 1 - 5 staged one context at a time (a barrier per context, the whole 60 / 110 KiB image again for every 64 pixels: at 16
// images per launch the d = 80 / 160 levels ran 8x / 11x their forward) or needed the forward AND backward images of all K + 2
// contexts in LDS at once (d <= 48 only). This kernel keeps, per (image, head), the BACKWARD operand image [KQ | VQ | KP] of as
// many contexts as fit the CU's 160 KiB — 29 KiB per context at d = 40 (all K + 2 up to K = 3), 45 at d = 80 (contexts 0, 1 and
// the first local one), 80 at d = 160 (contexts 0 and 1) — and reads the remaining local contexts, which only the 16-pixel groups a
// disc touches need, as MFMA operands straight from the packed image in L2 (one coalesced 1-KiB buffer load per fragment: the
// accessor idea of sta_xattn_proj.hip's locals-from-L2 forward). A workgroup = 4 waves x 16 pixels passes ONE barrier and then
// walks `p.iters` strided pixel tiles with that image, the waves running independently. The forward half [VP] of the packed image
// is never touched: with
//   dcoef_i = sum_px mask_i (delta_i - sum_key P_0 (dO1 V_0^T)),   delta_i = sum_key P_i (dO1 V_i^T)
// the attention outputs A_i, A_u are not needed (see the file header), and the upstream of context 0,
// dO0 - (sum_i coef_i mask_i) dO1, is formed in fp32 from the two dP^T products instead of being rounded to 16 bits.
//
// ONE WAVE PER SIMD, by construction (4 waves per workgroup, one workgroup per CU: the launch asks for more than half the LDS), one
// 16-pixel tile per wave: see launch_bwd_any for the wider builds that are not shipped and why.
constexpr int BWD_MAXIT = 16;
__host__ __device__ constexpr int bwd_frags(int ndt) { return 2 * NKT * nks_of(ndt) + NPS * ndt; }

// where a context's backward operands come from: an LDS slot [KQ | VQ | KP] or the full packed image [KQ | VP | VQ | KP] in L2
template <typename V8, int NKF>
struct LdsBwdFrags {
  const V8* base;                                  // slot + lane
  __device__ __forceinline__ V8 kq(int f) const { return base[f * 64]; }
  __device__ __forceinline__ V8 vq(int f) const { return base[(NKF + f) * 64]; }
  __device__ __forceinline__ V8 kp(int f) const { return base[(2 * NKF + f) * 64]; }
};
template <typename V8, int NKF, int NFWD>
struct SrdBwdFrags {
  __amdgpu_buffer_rsrc_t r;
  unsigned voff, soff;                             // lane * 16; byte offset of the (context, head) block (wave-uniform)
  __device__ __forceinline__ V8 kq(int f) const { return srd_load16<V8>(r, voff, soff + 1024u * (unsigned)f); }
  __device__ __forceinline__ V8 vq(int f) const { return srd_load16<V8>(r, voff, soff + 1024u * (unsigned)(NFWD + f)); }
  __device__ __forceinline__ V8 kp(int f) const { return srd_load16<V8>(r, voff, soff + 1024u * (unsigned)(NFWD + NKF + f)); }
};

// One context for the QT 16-pixel tiles of a wave: S^T and dP^T per key tile, softmax, delta, dS, dQ^T += KP dS^T. Every operand
// fragment that comes out of LDS (or L2) serves the QT tiles: with one wave per SIMD the second tile is what fills the first
// one's LDS round trips and MFMA dependencies, and it halves the LDS reads per MFMA.
// DUAL (context 0 where a disc touches the wave): a second walk over the VQ tiles with dO1 gives, tile by tile,
// du1 = sum_key P (VQ.dO1) and dP^T -= wsum (VQ.dO1) in fp32 — four accumulator registers instead of a third S^T-sized set.
// `gscale` is per lane (softmax scale x the pixel's blend weight); delta = sum_key P dP of the UNWEIGHTED upstream.
template <typename T, int NDT, int QT, bool FAST, bool DUAL, typename FR>
__device__ __forceinline__ void attend_bwd_res(const FR fr, const typename Tr<T>::V8 (&qf)[QT][nks_of(NDT)],
                                               const typename Tr<T>::V8 (&gf)[QT][nks_of(NDT)],
                                               const typename Tr<T>::V8 (&g1)[QT][nks_of(NDT)], const f32x4 kb4,
                                               const float sl2e, const int g, const int M, const float (&gscale)[QT],
                                               const float (&wsum)[QT], f32x4 (&dq)[QT][NDT], float (&delta)[QT], float (&du1)[QT]) {
  using V8 = typename Tr<T>::V8;
  constexpr int NKS = nks_of(NDT);
  // (fragments are requested per key tile / head-dim tile right in front of their MFMAs and left to the compiler's scheduler, which
  // hoists them as far as the registers allow: an explicit double buffer measured the same, 106.9 vs 105.0 us at level 0)
  f32x4 st[QT][NKT], dp[QT][NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    V8 kt[NKS], vt[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      kt[s] = fr.kq(t * NKS + s);
      vt[s] = fr.vq(t * NKS + s);
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 as = (FAST && t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 ap = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        as = Tr<T>::mfma(kt[s], qf[qt][s], as);
        ap = Tr<T>::mfma(vt[s], gf[qt][s], ap);
      }
      st[qt][t] = as;
      dp[qt][t] = ap;
    }
  }
  float inv[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) inv[qt] = FAST ? softmax_biased(st[qt], sl2e) : softmax_keys_fast(st[qt], g, M, sl2e);
  if constexpr (DUAL) {
    float a[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) a[qt] = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      V8 vt[NKS];
#pragma unroll
      for (int s = 0; s < NKS; ++s) vt[s] = fr.vq(t * NKS + s);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 ab = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NKS; ++s) ab = Tr<T>::mfma(vt[s], g1[qt][s], ab);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a[qt] = __builtin_fmaf(st[qt][t][r], ab[r], a[qt]);
          dp[qt][t][r] = __builtin_fmaf(-wsum[qt], ab[r], dp[qt][t][r]);
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) du1[qt] = bfly_sum(a[qt]) * inv[qt];
  }
  V8 pb[QT][NPS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float dl = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dl = __builtin_fmaf(st[qt][t][r], dp[qt][t][r], dl);
    delta[qt] = bfly_sum(dl) * inv[qt];
    const float sc = inv[qt] * gscale[qt];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[qt][t][r] = st[qt][t][r] * sc * (dp[qt][t][r] - delta[qt]);   // padded keys: st == 0
    tiles_to_b<T>(st[qt], pb[qt]);
  }
#pragma unroll
  for (int u = 0; u < NDT; ++u) {
    V8 kp[NPS];
#pragma unroll
    for (int s = 0; s < NPS; ++s) kp[s] = fr.kp(s * NDT + u);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int s = 0; s < NPS; ++s) dq[qt][u] = Tr<T>::mfma(kp[s], pb[qt][s], dq[qt][u]);
  }
}

template <typename T, int NDT, int NWV, int QT, bool FAST>
__global__ __launch_bounds__(64 * NWV, 1) void xattn_bwd_res_kernel(const Params pin) {
  using V8 = typename Tr<T>::V8;
  constexpr int NKS = nks_of(NDT);
  constexpr int NKF = NKT * NKS, NFWD = fwd_frags(NDT), NALL = all_frags(NDT);
  constexpr int SB = bwd_frags(NDT) * FRAG;        // bytes of one LDS slot
  constexpr int TP = 16 * NWV * QT;                // pixels per tile: NWV waves x QT sub-tiles x 16
  constexpr int CPT = TP / 64;                     // 64-pixel chunks per tile
  constexpr int MAXCH = BWD_MAXIT * CPT;
  // block -> (image, tile group, head): XCD-contiguous over the whole grid (see xattn_fwd_staged_kernel)
  const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
  const int Lg = xcd_remap(lin, (int)(gridDim.x * gridDim.y));
  const int img = Lg / (int)gridDim.x;
  const int L = Lg - img * (int)gridDim.x;
  const Params p = for_image<T, NDT>(pin, img, (size_t)pin.K * gridDim.x * NWV);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  int wt, h;
  if (p.H == 8) { wt = L >> 3; h = L & 7; } else { wt = L / p.H; h = L % p.H; }
  const int N = p.N, C = p.C, d = p.d, K = p.K, W = p.ntiles;
  const int mine = (p.tiles - wt + W - 1) / W;
  const int iters = mine < p.iters ? mine : p.iters;
  const int G = p.ntiles_aux;                      // LDS slots (>= 2): contexts 0, 1 and the first G - 2 active local ones

  // ---- prologue: weights, the mask bytes of every tile of this workgroup, contexts 0 and 1 -----------------------------
  const float coef_lane = p.coef[min(lane, K > 0 ? K - 1 : 0)];
  unsigned span_bits = 0;
  {
    unsigned tb[MAXCH];
#pragma unroll
    for (int j = 0; j < MAXCH; ++j)                // unconditional, clamped
      tb[j] = p.mask[min((wt + (j / CPT) * W) * TP + 64 * (j % CPT) + lane, N - 1)];
#pragma unroll
    for (int j = 0; j < MAXCH; ++j)
      span_bits |= (j / CPT < iters && (wt + (j / CPT) * W) * TP + 64 * (j % CPT) + lane < N) ? tb[j] : 0u;
  }
  const size_t ctx_stride = (size_t)p.H * NALL * FRAG;
  const char* img_h = p.packed + (size_t)h * NALL * FRAG;
  auto stage_ctx = [&](int c, int slot) {
    const char* src = img_h + (size_t)c * ctx_stride;
    char* dst = smem + (size_t)slot * SB;
    stage_frags(src, dst, NKF, wv, NWV, lane);                                       // KQ
    stage_frags(src + (size_t)NFWD * FRAG, dst + NKF * FRAG, NFWD, wv, NWV, lane);   // VQ | KP
  };
  stage_ctx(0, 0);
  stage_ctx(1, 1);
  unsigned tile_bits = 0;                          // discs that touch this workgroup's pixels (workgroup-uniform)
  span_bits &= (1u << K) - 1u;
  for (int i = 0; i < K; ++i)
    if (__ballot((span_bits >> i) & 1u)) tile_bits |= 1u << i;
  {
    unsigned bits = tile_bits;
    for (int n = 0; n < G - 2 && bits; ++n) {
      const int i = __builtin_ctz(bits);
      bits &= bits - 1;
      stage_ctx(2 + i, 2 + n);
    }
  }
  const unsigned row_bytes = (unsigned)C * (unsigned)sizeof(T);
  const unsigned act_bytes = 2u * (unsigned)N * row_bytes;
  const __amdgpu_buffer_rsrc_t q_srd = make_srd(p.q, act_bytes), g_srd = make_srd(p.dout, act_bytes);
  const __amdgpu_buffer_rsrc_t kv_srd = make_srd(img_h, (unsigned)(K + 2) * (unsigned)ctx_stride);
  const unsigned row1 = (unsigned)N * row_bytes;
  const f32x4 kb4 = last_tile_bias(g, p.M);
  const float sl2e = p.sl2e, scale = p.scale;
  float dc[MAXK];
#pragma unroll
  for (int i = 0; i < MAXK; ++i) dc[i] = 0.f;

  // B operands of a tile: 16 B per lane at head-dim offset 32 s + 8 g of the pixel's head row, both rows of q and of dO; pixels
  // >= N and offsets >= d are pushed out of the descriptor's range and read as zero. With ONE wave per SIMD nobody else covers the
  // HBM round trip of these loads, so the operands (and the mask byte) of tile it + 1 are requested before tile it is computed.
  struct TileOps {
    V8 q0[QT][NKS], g0[QT][NKS], g1[QT][NKS], q1[QT][NKS];
    unsigned own[QT];
  };
  auto request = [&](int it, TileOps& o) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int px = (wt + it * W) * TP + (wv * QT + qt) * 16 + c16;
      const bool ok = it < iters && px < N;
      o.own[qt] = p.mask[ok ? px : 0];
      const unsigned base = ok ? (unsigned)px * row_bytes + (unsigned)(h * d + 8 * g) * (unsigned)sizeof(T) : 0xfffffff0u;
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const unsigned vo = (32 * s + 8 * g < d) ? base : 0xfffffff0u;
        o.q0[qt][s] = srd_load16<V8>(q_srd, vo, 64u * s);
        o.g0[qt][s] = srd_load16<V8>(g_srd, vo, 64u * s);
        o.g1[qt][s] = srd_load16<V8>(g_srd, vo, row1 + 64u * s);
        o.q1[qt][s] = srd_load16<V8>(q_srd, vo, row1 + 64u * s);
      }
    }
  };
  TileOps cur, nxt;
  request(0, cur);
  wait_dma_and_sync();

  for (int it = 0; it < iters; ++it) {
    request(it + 1, nxt);
    bool valid[QT];
    unsigned ownbits[QT];
    T* dqbase[QT];
    float wsum[QT], gs[QT], zero[QT], delta[QT], du1[QT], unused[QT];
    unsigned anybits = 0;                          // discs that touch this wave's pixels (wave-uniform)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int px = (wt + it * W) * TP + (wv * QT + qt) * 16 + c16;
      valid[qt] = px < N;
      dqbase[qt] = (T*)p.out + (size_t)(valid[qt] ? px : 0) * C + h * d;
      ownbits[qt] = valid[qt] ? (cur.own[qt] & tile_bits) : 0u;
      wsum[qt] = 0.f;
      gs[qt] = scale;
      zero[qt] = 0.f;
      du1[qt] = 0.f;
      for (unsigned bits = tile_bits; bits; bits &= bits - 1) {
        const int i = __builtin_ctz(bits);
        const float ci = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), i));
        if (__ballot((ownbits[qt] >> i) & 1u)) anybits |= 1u << i;
        wsum[qt] += ((ownbits[qt] >> i) & 1u) ? ci : 0.f;
      }
    }
    f32x4 dq[QT][NDT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int u = 0; u < NDT; ++u) dq[qt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    // row 0: context 0 under dO0 - (sum_i coef_i mask_i) dO1
    const LdsBwdFrags<V8, NKF> f0{(const V8*)smem + lane};
    if (anybits) attend_bwd_res<T, NDT, QT, FAST, true>(f0, cur.q0, cur.g0, cur.g1, kb4, sl2e, g, p.M, gs, wsum, dq, delta, du1);
    else attend_bwd_res<T, NDT, QT, FAST, false>(f0, cur.q0, cur.g0, cur.g0, kb4, sl2e, g, p.M, gs, zero, dq, delta, unused);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (valid[qt]) store_row16<T, NDT>(dqbase[qt], dq[qt], g, d);
#pragma unroll
      for (int u = 0; u < NDT; ++u) dq[qt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // row 1: the global prompt, then the local prompts whose disc touches the wave
    const LdsBwdFrags<V8, NKF> f1{(const V8*)(smem + SB) + lane};
    attend_bwd_res<T, NDT, QT, FAST, false>(f1, cur.q1, cur.g1, cur.g1, kb4, sl2e, g, p.M, gs, zero, dq, delta, unused);
    for (unsigned bits = anybits; bits; bits &= bits - 1) {
      const int i = __builtin_ctz(bits);
      const int rank = __builtin_popcount(tile_bits & ((1u << i) - 1u));
      const float ci = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), i));
      float gl[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) gl[qt] = ((ownbits[qt] >> i) & 1u) ? scale * ci : 0.f;
      if (rank < G - 2) {
        const LdsBwdFrags<V8, NKF> fl{(const V8*)(smem + (size_t)(2 + rank) * SB) + lane};
        attend_bwd_res<T, NDT, QT, FAST, false>(fl, cur.q1, cur.g1, cur.g1, kb4, sl2e, g, p.M, gl, zero, dq, delta, unused);
      } else {
        const SrdBwdFrags<V8, NKF, NFWD> fl{kv_srd, (unsigned)lane * 16u, (unsigned)(2 + i) * (unsigned)ctx_stride};
        attend_bwd_res<T, NDT, QT, FAST, false>(fl, cur.q1, cur.g1, cur.g1, kb4, sl2e, g, p.M, gl, zero, dq, delta, unused);
      }
      float part = 0.f;                            // the sums are replicated over the four lane rows: row 0 speaks
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) part += (g == 0 && ((ownbits[qt] >> i) & 1u)) ? delta[qt] - du1[qt] : 0.f;
#pragma unroll
      for (int j = 0; j < MAXK; ++j) dc[j] += (j == i) ? part : 0.f;
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
      if (valid[qt]) store_row16<T, NDT>(dqbase[qt] + (size_t)N * C, dq[qt], g, d);
    cur = nxt;
  }
  // per-wave dcoef partials -> workspace [K][gridDim.x * NWV] (fixed slot per wave: deterministic)
#pragma unroll
  for (int i = 0; i < MAXK; ++i) {
    if (i < K) {
      float v = dc[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) p.aux[(size_t)i * gridDim.x * NWV + (size_t)L * NWV + wv] = v;
    }
  }
}

// Launch geometry of the LDS-resident backward: slots by LDS capacity, one workgroup per CU, enough strided tiles per
// workgroup that one round of workgroups covers the launch (at most BWD_MAXIT).
template <typename T, int NDT, int NWV, int QT>
int launch_bwd_res(const Params& p0, float* dcoef, hipStream_t st) {
  constexpr int SB = bwd_frags(NDT) * FRAG;
  constexpr int TP = 16 * NWV * QT;
  static_assert(2 * SB <= 160 * 1024, "contexts 0 and 1 must fit");
  Params p = p0;
  int G = (160 * 1024) / SB;
  if (G > p.K + 2) G = p.K + 2;
  if (const int v = g_sta_opt[STA_OPT_BWD_SLOTS]) { if (v >= 2 && v < G) G = v; }
  p.ntiles_aux = G;
  // more than half of the CU's LDS whatever the image needs: never a second workgroup (= a second wave per SIMD) beside this one
  const int lds = G * SB > 81 * 1024 ? G * SB : 81 * 1024;
  p.tiles = (p.N + TP - 1) / TP;
  long wg_per_head = 256L / ((long)p.H * p.n_img);
  if (wg_per_head < 1) wg_per_head = 1;
  if (wg_per_head > p.tiles) wg_per_head = p.tiles;
  int iters = (int)((p.tiles + wg_per_head - 1) / wg_per_head);
  if (iters > BWD_MAXIT) iters = BWD_MAXIT;
  if (const int v = g_sta_opt[STA_OPT_STAGED_TILES]) iters = v < BWD_MAXIT ? v : BWD_MAXIT;
  if (iters > p.tiles) iters = p.tiles;
  p.iters = iters;
  p.ntiles = (p.tiles + iters - 1) / iters;                 // workgroups per head
  const int nwg = p.ntiles * p.H;
  auto launch = [&](auto kernel, StaLdsAttr& attr) {
    if (!attr.ensure((const void*)kernel, 160 * 1024)) return fail(STA_E_LAUNCH, "hipFuncSetAttribute(bwd resident) failed");
    hipLaunchKernelGGL(kernel, dim3(nwg, p.n_img), dim3(64 * NWV), lds, st, p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? STA_OK : fail(STA_E_LAUNCH, "bwd resident launch: %s", hipGetErrorString(e));
  };
  static StaLdsAttr attr_fast, attr_any;
  const int rc = p.M > 16 * (NKT - 1) ? launch(xattn_bwd_res_kernel<T, NDT, NWV, QT, true>, attr_fast)
                                      : launch(xattn_bwd_res_kernel<T, NDT, NWV, QT, false>, attr_any);
  if (rc) return rc;
  if (p.K > 0) {
    hipLaunchKernelGGL(dcoef_reduce_kernel, dim3(p.K, p.n_img), dim3(256), 0, st, p.aux, dcoef, nwg * NWV);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(STA_E_LAUNCH, "dcoef reduce launch: %s", hipGetErrorString(e));
  }
  return STA_OK;
}

// The product launches 4 waves x ONE 16-pixel tile per wave, one workgroup per CU. The two wider builds — eight waves (two per SIMD),
// or two tiles per wave — are 14 - 27 % faster at level 0 (101 / 86 vs 117 us at 16 images) and are NOT shipped: both give a handful
// of wrong values per launch, at ONE dS element of the tile (lane row 3, accumulator register 2 of one key tile: key 30 in one
// build, key 62 in the other), different pixels every run in the eight-wave build. profiles/r06_bwd_race.md holds what was ruled out
// (every pair of the measured hazard table, waits, the packed-fp32 chain replayed alone, LDS-return and chained-MFMA probes) and
// tools/dbg/ the scripts; -DSTA_EXPERIMENT_BWD_WIDE compiles them in for that work (STA_OPT_BWD_WAVES = 8, STA_OPT_STAGED_QT = 2).
template <typename T, int NDT>
int launch_bwd_any(const Params& p, float* dcoef, hipStream_t st) {
#ifdef STA_EXPERIMENT_BWD_WIDE
  if (g_sta_opt[STA_OPT_BWD_WAVES] == 8) return launch_bwd_res<T, NDT, 8, 1>(p, dcoef, st);
  if constexpr (NDT <= 6) {
    if (g_sta_opt[STA_OPT_STAGED_QT] == 2) return launch_bwd_res<T, NDT, 4, 2>(p, dcoef, st);
  }
#endif
  return launch_bwd_res<T, NDT, 4, 1>(p, dcoef, st);
}

template <typename T>
int dispatch_bwd(const Params& p, float* dcoef, hipStream_t st) {
  switch ((p.d + 15) / 16) {
    case 1: return launch_bwd_any<T, 1>(p, dcoef, st);
    case 2: return launch_bwd_any<T, 2>(p, dcoef, st);
    case 3: return launch_bwd_any<T, 3>(p, dcoef, st);
    case 4: return launch_bwd_any<T, 4>(p, dcoef, st);
    case 5: return launch_bwd_any<T, 5>(p, dcoef, st);
    case 6: return launch_bwd_any<T, 6>(p, dcoef, st);
    case 7: return launch_bwd_any<T, 7>(p, dcoef, st);
    case 8: return launch_bwd_any<T, 8>(p, dcoef, st);
    case 9: return launch_bwd_any<T, 9>(p, dcoef, st);
    case 10: return launch_bwd_any<T, 10>(p, dcoef, st);
  }
  return fail(STA_E_UNSUP, "head dim %d unsupported", p.d);
}

}  // namespace

extern "C" {

size_t sta_xattn_bwd_workspace_bytes(int n_img, int N, int heads, int K) {
  if (n_img <= 0 || N <= 0 || heads <= 0 || K <= 0) return 16;
  // one float per (object, wave): 16-pixel wave tiles rounded up to whole workgroups
  return (size_t)n_img * K * ((N + 15) / 16 + 16) * heads * sizeof(float);
}

int sta_xattn_bwd(const void* q, const void* packed, const uint8_t* mask, const float* coef,
                  const void* dout, void* dq, float* dcoef, void* workspace, int n_img, int N, int C, int heads,
                  int M, int K, float scale, int dtype, void* stream) {
  g_err[0] = 0;
  if (!q || !packed || !dout || !dq) return fail(STA_E_ARG, "null pointer");
  if (n_img < 1 || n_img > 65535) return fail(STA_E_ARG, "n_img=%d", n_img);
  if (int rc = check_shape(N, C, heads, M, K)) return rc;
  if (K > 0 && (!mask || !coef || !dcoef || !workspace)) return fail(STA_E_ARG, "mask/coef/dcoef/workspace required when K > 0");
  if (dtype != STA_BF16 && dtype != STA_F16) return fail(STA_E_UNSUP, "dtype %d", dtype);
  if ((size_t)2 * N * C * 2 >= 0xfffffff0ull) return fail(STA_E_UNSUP, "one image of activations must stay below 4 GiB");
  Params p{};
  p.q = q; p.packed = (const char*)packed; p.mask = mask; p.coef = coef; p.out = dq; p.dout = dout;
  if (K == 0) {  // unconditional prologue loads: readable (ignored) bytes
    p.mask = (const uint8_t*)q;
    p.coef = (const float*)q;
  }
  p.aux = (float*)workspace; p.N = N; p.C = C; p.H = heads; p.d = C / heads; p.M = M; p.K = K; p.n_img = n_img;
  p.scale = scale; p.sl2e = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  return dtype == STA_BF16 ? dispatch_bwd<__bf16>(p, dcoef, st) : dispatch_bwd<_Float16>(p, dcoef, st);
}

}  // extern "C"

