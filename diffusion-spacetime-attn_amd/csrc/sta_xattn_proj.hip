// sta_xattn_proj.hip — spatial-temporal cross-attention forward WITH the query projection inside
// (gfx950 only). C-ABI: sta_xattn_pack_wq / sta_xattn_pack_kv_proj / sta_xattn_fwd_proj.
//
// What it replaces (reference, attention_optimization/stable-diffusion/ldm/modules/attention.py):
//   :178      q = self.to_q(x)                       (x = norm2(hidden), recomputed K+1 times there)
//   :175-197  CrossAttention.forward for the K+2 (row, context) pairs that reach the output
//   :278-294  the disc-masked, coef-weighted global/local blend
// i.e. sta_xattn_fwd plus the GEMM in front of it: the [2][N][C] query tensor never exists in HBM.
//
// Why: the attention alone has 154 flop per HBM byte (below the 310 flop/B ridge of an MI355X) and its
// instruction stream is VALU bound (softmax: ~9 VALU per MFMA, matrix pipe 14 % busy). The projection is pure
// MFMA work on the SAME pixels: 2*C flop per activation byte more, no extra HBM traffic (y in instead of q in),
// and its MFMAs issue in the shadow of the other waves' softmax VALU.
//
// Everything stays "swapped" (pixel = MFMA column, see sta_xattn.hip):
//   Q^T[dd][px] = Wq_h[dd][:] . y[px][:]     A = Wq fragments (LDS), B = y rows (global, 16 B per lane)
// The accumulator of head-dim tile u holds Q^T[16u + 4g + r][px = c]; packed as 16-bit pairs it IS the B
// operand of S^T = K.Q^T once the head-dim axis of the K fragments is permuted at pack time
// (k-slot 8g+j of step s  <->  head dim 32s + 16(j>>2) + 4g + (j&3), the same permutation the PV product uses
// for its keys): no LDS round trip and no cross-lane traffic between the projection and the attention.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sta_xattn.h"
#include "sta_internal.h"
#include "sta_xattn_dev.h"
#include "sta_xattn_proj3.h"

namespace {

#ifndef STA_PROJ_ABLATE
#define STA_PROJ_ABLATE 0   // timing experiments only (tools/asm_patch_ab.py flag:-DSTA_PROJ_ABLATE=bits; wrong results): 1 no output stores,
#endif                      // 2 no y refills in the loop, 4 no attention, 8 no projection MFMAs, 16 no local contexts, 32 no Wq fragment reads

constexpr int RING_MIN = 5;   // k-steps (32 channels each) of y per ring refill; C % (32 * RING_MIN) == 0

// to_q.weight [C][C] (row = output channel h*d + dd, col = input channel) -> per head [s][u] fragments:
// lane (g, c) of fragment (s, u) holds Wq[h*d + 16u + c][32s + 8g .. +7]; rows 16u + c >= d are zero.
template <typename T>
__global__ __launch_bounds__(64) void pack_wq_kernel(const T* __restrict__ wq, T* __restrict__ packed, int C, int d,
                                                     int ndt) {
  const int f = blockIdx.x, h = blockIdx.y;
  const int s = f / ndt, u = f % ndt;
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int row = 16 * u + c;
  typename Tr<T>::V8 val;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = 32 * s + 8 * g + j;
    val[j] = (row < d && col < C) ? wq[((size_t)h * d + row) * C + col] : (T)0.0f;
  }
  *((typename Tr<T>::V8*)((char*)packed + ((size_t)h * gridDim.x + f) * FRAG) + lane) = val;
}

// K,V [n_ctx][M][C] -> forward-only fragment image [ctx][head][KQ' | VP]; KQ' = the K rows with the head-dim
// axis in projected-query order (see the file header), VP exactly as in sta_xattn.hip (incl. the ones row).
template <typename T>
__global__ __launch_bounds__(64) void pack_kv_proj_kernel(const T* __restrict__ k, const T* __restrict__ v,
                                                          T* __restrict__ packed, int M, int C, int H, int d, int ndt) {
  const int nks = nks_of(ndt);
  const int f = blockIdx.x;               // 0 .. fwd_frags(ndt) - 1
  const int ch = blockIdx.y;              // ctx * H + h
  const int ctx = ch / H, h = ch % H;
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  typename Tr<T>::V8 val;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    T x = (T)0.0f;
    if (f < NKT * nks) {
      const int t = f / nks, s = f % nks;
      const int key = 16 * t + c, dd = pv_key(s, g, j);      // head dim held by k-slot (g, j) of step s
      if (key < M && dd < d) x = k[((size_t)ctx * M + key) * C + h * d + dd];
    } else {
      const int f2 = f - NKT * nks;
      const int s = f2 / ndt, u = f2 % ndt;
      const int key = pv_key(s, g, j), dd = 16 * u + c;
      if (key < M && dd < d) x = v[((size_t)ctx * M + key) * C + h * d + dd];
      else if (key < M && dd == d) x = (T)1.0f;             // ones row: softmax denominator out of the PV MFMAs
    }
    val[j] = x;
  }
  *((typename Tr<T>::V8*)((char*)packed + ((size_t)ch * gridDim.x + f) * FRAG) + lane) = val;
}

struct PParams {
  const void* y;         // [I][2][N][C]  norm2(hidden)
  const char* wq;        // packed to_q.weight: [H][nkc][NDT] fragments
  const char* kv;        // forward-only K/V image: [I][K+2][H][KQ' | VP]
  const uint8_t* mask;   // [I][N] bit field
  const float* coef;     // [I][K]
  void* out;             // [I][2][N][C]
  int N, C, H, d, M, K;
  int nkc;               // k-steps of the projection: C / 32
  int W;                 // workgroups per (head, image)
  int tiles;             // pixel tiles per head
  int iters;             // pixel tiles a workgroup walks at most
  float sl2e;
};

// One workgroup = NWV waves x 16 pixels, one head, one image. It copies the head's Wq slice and the fragments of
// ALL K+2 contexts into LDS once (LDS-DMA; 30 + 19 (K+2) KiB at d = 40), passes one barrier and then walks
// `iters` strided pixel tiles: per tile and wave
//   projection  : NDT * nkc * 2 MFMAs (both batch rows share every Wq fragment read), y rows streamed through a
//                 RING-deep register ring: when k-step s is consumed its slot is refilled with step s + RING — of
//                 this tile, or, past its end, of the NEXT tile, whose rows therefore land under this tile's
//                 attention. The y rows are re-read by the 8 head workgroups of a tile group, i.e. 8x the activation
//                 bytes come through the CU's vector-memory path (from L2): the ring depth is what keeps that path
//                 busy — 2 x RING loads of 1 KiB in flight per wave at all times. That path, not the MFMA or VALU
//                 pipes, bounds the kernel: 671 MB of y per 16-image launch at level 0 arrive at 7.6 TB/s
//                 (profiles/r02_proj_fusion.md)
//   attention   : contexts 0, 1 and the local contexts whose disc touches the wave's pixels, from LDS
//                 (attend_staged, shared with sta_xattn.hip) — blend in registers, 16-byte stores.
// No barrier after the prologue: waves drift apart, one wave's projection MFMAs run beside another's softmax VALU.
//
// LL2 (locals from L2; SD-v1 level 1: C = 640, d = 80): the Wq slice (100 KiB) and the TWO mandatory contexts (2 x 30 KiB) fill
// the CU's 160 KiB exactly; the local contexts — needed only by the 16-pixel groups a disc touches — are read as MFMA operands
// straight from the packed image in global memory (one coalesced 1-KiB buffer load per fragment, L2 hits: the image of a prompt
// is 30 KiB per (context, head) and every workgroup of the image walks it).
// YFRAG: y in query-fragment order (sta_add_layernorm_qfrag), as in the head-pair kernel: one load = one contiguous KiB.
// (A fully unrolled projection loop with the Wq fragments of k-step s + 1 requested before the MFMAs of step s — the schedule of
// sta_xattn_proj3.hip — measured 3 % SLOWER here than hipcc's just-in-time reads at 36 registers more: profiles/r05_level1_proj.md.)
template <typename T, int NDT, int NWV, int RING, bool LL2 = false, bool YFRAG = false>
__global__ __launch_bounds__(64 * NWV, 2) void xattn_fwd_proj_kernel(const PParams p) {
  using V8 = typename Tr<T>::V8;
  constexpr unsigned YSTEP = YFRAG ? 1024u : 64u;      // bytes between consecutive k-steps of a lane's y loads
  constexpr int NKS = nks_of(NDT);
  constexpr int NFWD = fwd_frags(NDT);
  constexpr int CB = NFWD * FRAG;                 // bytes of one staged context
  constexpr int TP = 16 * NWV;                    // pixels per tile
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  // block -> (image, tile group, head): XCD-contiguous over the whole grid, so that the 8 heads of a tile group —
  // which read the SAME y rows — share one L2 (y is fetched from HBM once, the other 7 reads are L2 hits)
  const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
  const int Lg = xcd_remap(lin, (int)(gridDim.x * gridDim.y));
  const int img = Lg / (int)gridDim.x;
  const int L = Lg - img * (int)gridDim.x;
  int wt, h;
  if (p.H == 8) { wt = L >> 3; h = L & 7; } else { wt = L / p.H; h = L % p.H; }
  const int N = p.N, C = p.C, d = p.d, K = p.K, nkc = p.nkc, W = p.W;
  const unsigned row_bytes = (unsigned)C * (unsigned)sizeof(T);
  const size_t act = (size_t)2 * N * row_bytes;
  const char* yb = (const char*)p.y + img * act;
  T* ob = (T*)((char*)p.out + img * act);
  const size_t ctx_stride = (size_t)p.H * CB;
  const char* kv = p.kv + ((size_t)img * (K + 2) * p.H + h) * CB;
  const uint8_t* mask = p.mask + (size_t)img * N;
  const float coef_lane = p.coef[(size_t)img * K + min(lane, K > 0 ? K - 1 : 0)];
  const int nwq = NDT * nkc;                      // Wq fragments of this head
  char* lds_ctx = smem + (size_t)nwq * FRAG;

  // ---- prologue: LDS-DMA of Wq_h and every context, first y steps of the first tile -------------------------
  stage_frags(p.wq + (size_t)h * nwq * FRAG, smem, nwq, wv, NWV, lane);
  for (int c = 0; c < (LL2 ? 2 : K + 2); ++c) stage_frags(kv + c * ctx_stride, lds_ctx + c * CB, NFWD, wv, NWV, lane);
  // LL2: descriptor over this image's whole packed K/V image; block (ctx, h) sits at (ctx * H + h) * CB
  SrdFrags<V8> gfr;
  gfr.r = make_srd(p.kv + (size_t)img * (K + 2) * ctx_stride, (unsigned)((K + 2) * ctx_stride));
  gfr.voff = (unsigned)lane * 16u;
  gfr.soff = 0u;

  const int mine = (p.tiles - wt + W - 1) / W;    // tiles wt, wt + W, ... of this workgroup
  const int iters = mine < p.iters ? mine : p.iters;
  const __amdgpu_buffer_rsrc_t y_srd = make_srd(yb, (unsigned)act);
  const unsigned row1 = (unsigned)N * row_bytes;
  // pixels >= N (and tiles past the end) are pushed out of the descriptor's range: they read as zeros
  // every head walks its tiles in the same order: the 8 head workgroups of a tile group then touch the same y rows
  // at about the same time and 7 of the 8 reads hit L2. (Starting head h a few tiles ahead, to spread the requests
  // over L2 channels, measured equal to slower: 87.3 / 91.5 / 101 us with heads 4 / 2 / 1 per phase vs 87.2 us.)
  auto tile_of = [&](int it) -> int { return wt + it * W; };
  auto voff_of = [&](int it) -> unsigned {
    if constexpr (YFRAG) {      // fragment s of the wave's 16-pixel group: byte offset of its first pixel's row + 1024 s + 16 lane
      const int px0 = tile_of(it) * TP + wv * 16;
      return (it < iters && px0 < N) ? (unsigned)px0 * row_bytes + (unsigned)lane * 16u : 0xfffffff0u;
    }
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
    yr0[j] = srd_load16<V8>(y_srd, voff, YSTEP * j);
    yr1[j] = srd_load16<V8>(y_srd, voff, row1 + YSTEP * j);
  }
  const f32x4 kb4 = last_tile_bias(g, p.M);
  const float sl2e = p.sl2e;
  const int sumrow = (d & 15) ? (d & 15) : -1;
  const unsigned kmask = (1u << K) - 1u;
  STA_T_INIT();
  STA_T(0);
  wait_dma_and_sync();
  STA_T(1);

  for (int it = 0; it < iters; ++it) {
    if (it == 1) STA_T(2);
    // ---- projection: Q^T tiles of both batch rows --------------------------------------------------------
    f32x4 qa0[NDT], qa1[NDT];
#pragma unroll
    for (int u = 0; u < NDT; ++u) {
      qa0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      qa1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const V8* wf = (const V8*)smem + lane;
    for (int s0 = 0; s0 < nkc; s0 += RING) {
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        const int s = s0 + j;
        V8 a[NDT];
#pragma unroll
        for (int u = 0; u < NDT; ++u) a[u] = wf[((STA_PROJ_ABLATE & 32 ? 0 : s) * NDT + u) * 64];
#pragma unroll
        for (int u = 0; u < NDT; ++u) {
#if STA_PROJ_ABLATE & 8
          asm volatile("" : "+v"(qa0[u]), "+v"(qa1[u]) : "v"(a[u]), "v"(yr0[j]), "v"(yr1[j]));
#else
          qa0[u] = Tr<T>::mfma(a[u], yr0[j], qa0[u]);
          qa1[u] = Tr<T>::mfma(a[u], yr1[j], qa1[u]);
#endif
        }
        // refill this slot with step s + RING: of this tile, or of the next one (zeros past the last tile)
        const bool wrap = s + RING >= nkc;        // scalar
        const unsigned vo = wrap ? voffn : voff;
        const unsigned so = YSTEP * (unsigned)(wrap ? s + RING - nkc : s + RING);
#if !(STA_PROJ_ABLATE & 2)
        yr0[j] = srd_load16<V8>(y_srd, vo, so);
        yr1[j] = srd_load16<V8>(y_srd, vo, row1 + so);
#endif
      }
    }
    if (it == 1) STA_T(3);
    const int px_own = tile_of(it) * TP + wv * 16 + c16;
    const bool valid = px_own < N;
    voff = voffn;
    voffn = voff_of(it + 2);
    const unsigned mbn = mask_of(it + 1);
    // accumulators -> B operands of S^T (rounded to T once, as a GEMM epilogue would)
    V8 q0[1][NKS], q1[1][NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = 2 * s + (j >> 2);
        q0[0][s][j] = (t < NDT) ? (T)qa0[t][j & 3] : (T)0.0f;
        q1[0][s][j] = (t < NDT) ? (T)qa1[t][j & 3] : (T)0.0f;
      }

    // ---- attention + blend -------------------------------------------------------------------------------
    f32x4 au[1][NDT], ac[1][NDT];
    float w[1] = {0.f};
#if STA_PROJ_ABLATE & 4
#pragma unroll
    for (int u = 0; u < NDT; ++u) { au[0][u] = qa0[u]; ac[0][u] = qa1[u]; }
    asm volatile("" :: "v"(q0[0][0]), "v"(q1[0][0]));
#else
    attend_staged<T, NDT, 1, 0>((const V8*)lds_ctx + lane, q0, kb4, sl2e, w, au, ac, sumrow);
    if (it == 1) STA_T(4);
    attend_staged<T, NDT, 1, 1>((const V8*)(lds_ctx + CB) + lane, q1, kb4, sl2e, w, au, ac, sumrow);
    if (it == 1) STA_T(5);
    const unsigned mbits = valid ? (mb & kmask) : 0u;
    for (int i = 0; i < ((STA_PROJ_ABLATE & 16) ? 0 : K); ++i) {
      if (!__ballot((mbits >> i) & 1u)) continue;  // none of this wave's pixels inside disc i
      const float cw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), i));
      w[0] = ((mbits >> i) & 1u) ? cw : 0.f;
      if constexpr (LL2) {
        gfr.soff = (unsigned)(((2 + i) * p.H + h) * CB);
        attend_staged<T, NDT, 1, 2>(gfr, q1, kb4, sl2e, w, au, ac, sumrow);
      } else {
        attend_staged<T, NDT, 1, 2>((const V8*)(lds_ctx + (size_t)(2 + i) * CB) + lane, q1, kb4, sl2e, w, au, ac, sumrow);
      }
    }
#endif
    if (it == 1) STA_T(6);
#if STA_PROJ_ABLATE & 1
#pragma unroll
    for (int u = 0; u < NDT; ++u) asm volatile("" :: "v"(au[0][u]), "v"(ac[0][u]));      // the results stay alive without their stores
#endif
    if (valid && !(STA_PROJ_ABLATE & 1)) {
      T* obase = ob + (size_t)px_own * C + h * d;
      store_row16<T, NDT>(obase, au[0], g, d);
      store_row16<T, NDT>(obase + (size_t)N * C, ac[0], g, d);
    }
    if (it == 1) STA_T(7);
    mb = mbn;
  }
  STA_T(8);
  STA_T_END();
}

// --------------------------------------------------------------------------------------------------------------------------------
// The same launch for head dims whose Wq slice does NOT fit a CU (round 6; SD-v1 levels 2 and mid: C = 1280, d = 160: 400 KiB per head):
// Wq is STREAMED — per pixel tile its 40 k-steps pass through a two-slot LDS ring in chunks of two k-steps (20 1-KiB fragments, the
// pattern of csrc/sta_rowgemm.hip::to_out_ln_ofrag_kernel) — beside the two mandatory contexts (2 x 55 KiB resident); the local
// contexts come from L2 as in the LL2 variant above. q never exists in HBM at any level with this.
//   per chunk and wave: 2 k-steps x NDT tiles x 2 batch rows = 40 MFMAs behind 20 fragment reads, one barrier;
//   the DMA of chunk c + 1 is issued before the MFMAs of chunk c; `s_waitcnt vmcnt(4)` at the end of a chunk leaves exactly the four
//   y-ring refills of the chunk in flight (the ring is what keeps the vector-memory path busy) and covers the wave's three DMA pieces;
//   the ring runs THROUGH the attention phase: the last chunk of a tile requests chunk 0 again (of the next tile; wasted once at the
//   end of a workgroup's life), so every chunk iteration issues the same loads and the wait count is a constant.
// Traffic: a workgroup re-reads the head's 400 KiB per 128-pixel tile from L2 (level 2 at 64 images: 1024 tiles = 410 MB per launch)
// against the 2 x 84 MB HBM round trip of q and a library GEMM launch it replaces.
constexpr int WQS_CH = 2;        // k-steps per Wq chunk
constexpr int WQS_RING = 4;      // y k-steps in flight per batch row (divides nkc together with WQS_CH: C % 128 == 0)
template <typename T, int NDT, int NWV, int QT>
__global__ __launch_bounds__(64 * NWV, 1) void xattn_fwd_proj_wqs_kernel(const PParams p) {
  using V8 = typename Tr<T>::V8;
  constexpr int RING = WQS_RING, CH = WQS_CH;
  constexpr int NKS = nks_of(NDT);
  constexpr int NFWD = fwd_frags(NDT);
  constexpr int CB = NFWD * FRAG;                 // bytes of one staged context
  constexpr int TP = 16 * NWV * QT;               // pixels per tile: NWV waves x QT sub-tiles x 16
  constexpr int CF = CH * NDT;                    // fragments per chunk
  constexpr int PIECES = (CF + NWV - 1) / NWV;    // DMA pieces per wave and chunk (the last ones may repeat a fragment: same bytes, same place)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
  const int Lg = xcd_remap(lin, (int)(gridDim.x * gridDim.y));
  const int img = Lg / (int)gridDim.x;
  const int L = Lg - img * (int)gridDim.x;
  int wt, h;
  if (p.H == 8) { wt = L >> 3; h = L & 7; } else { wt = L / p.H; h = L % p.H; }
  const int N = p.N, C = p.C, d = p.d, K = p.K, nkc = p.nkc, W = p.W;
  const int NC = nkc / CH;                        // chunks per tile
  const unsigned row_bytes = (unsigned)C * (unsigned)sizeof(T);
  const size_t act = (size_t)2 * N * row_bytes;
  const char* yb = (const char*)p.y + img * act;
  T* ob = (T*)((char*)p.out + img * act);
  const size_t ctx_stride = (size_t)p.H * CB;
  const char* kv = p.kv + ((size_t)img * (K + 2) * p.H + h) * CB;
  const uint8_t* mask = p.mask + (size_t)img * N;
  const float coef_lane = p.coef[(size_t)img * K + min(lane, K > 0 ? K - 1 : 0)];
  const char* wq_h = p.wq + (size_t)h * NDT * nkc * FRAG;
  char* lds_ring = smem;                          // 2 slots x CF fragments
  char* lds_ctx = smem + 2 * CF * FRAG;

  auto stage_chunk = [&](int c, int slot) {       // PIECES pieces per wave, always (constant wait counts)
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      int f = wv + NWV * i;
      f = f < CF ? f : CF - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wq_h + ((size_t)c * CF + f) * FRAG + lane * 16),
                                       (__attribute__((address_space(3))) void*)(lds_ring + (slot * CF + f) * FRAG), 16, 0, 0);
    }
  };
  stage_chunk(0, 0);
  for (int c = 0; c < 2; ++c) stage_frags(kv + c * ctx_stride, lds_ctx + c * CB, NFWD, wv, NWV, lane);
  SrdFrags<V8> gfr;
  gfr.r = make_srd(p.kv + (size_t)img * (K + 2) * ctx_stride, (unsigned)((K + 2) * ctx_stride));
  gfr.voff = (unsigned)lane * 16u;
  gfr.soff = 0u;

  const int mine = (p.tiles - wt + W - 1) / W;
  const int iters = mine < p.iters ? mine : p.iters;
  const __amdgpu_buffer_rsrc_t y_srd = make_srd(yb, (unsigned)act);
  const unsigned row1 = (unsigned)N * row_bytes;
  auto px_of = [&](int it, int qt) -> int { return (wt + it * W) * TP + (wv * QT + qt) * 16 + c16; };
  auto voff_of = [&](int it, int qt) -> unsigned {
    const int px = px_of(it, qt);
    return (it < iters && px < N) ? (unsigned)px * row_bytes + (unsigned)g * 16u : 0xfffffff0u;
  };
  auto mask_of = [&](int it, int qt) -> unsigned {
    const int px = px_of(it, qt);
    return mask[(it < iters && px < N) ? px : 0];
  };
  V8 yr0[QT][RING], yr1[QT][RING];
  unsigned voff[QT], voffn[QT], mb[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    voff[qt] = voff_of(0, qt);
    voffn[qt] = voff_of(1, qt);
    mb[qt] = mask_of(0, qt);
#pragma unroll
    for (int j = 0; j < RING; ++j) {
      yr0[qt][j] = srd_load16<V8>(y_srd, voff[qt], 64u * j);
      yr1[qt][j] = srd_load16<V8>(y_srd, voff[qt], row1 + 64u * j);
    }
  }
  const f32x4 kb4 = last_tile_bias(g, p.M);
  const float sl2e = p.sl2e;
  const int sumrow = (d & 15) ? (d & 15) : -1;
  const unsigned kmask = (1u << K) - 1u;
  wait_dma_and_sync();

  for (int it = 0; it < iters; ++it) {
    f32x4 qa0[QT][NDT], qa1[QT][NDT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int u = 0; u < NDT; ++u) {
        qa0[qt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        qa1[qt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    const V8* wf = (const V8*)lds_ring + lane;
    // RING k-steps = RING / CH chunks per trip: ring slots and LDS slots have compile-time indices
    for (int s0 = 0; s0 < nkc; s0 += RING) {
#pragma unroll
      for (int cc = 0; cc < RING / CH; ++cc) {
        const int c = s0 / CH + cc;               // this tile's chunk, in LDS slot cc & 1 (RING / CH is even)
        stage_chunk(c + 1 < NC ? c + 1 : 0, (cc + 1) & 1);
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
          const int j = cc * CH + jj, s = s0 + j;
#pragma unroll
          for (int u = 0; u < NDT; ++u) {
            const V8 a = wf[(((cc & 1) * CH + jj) * NDT + u) * 64];      // one fragment read serves 2 QT MFMAs
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
              qa0[qt][u] = Tr<T>::mfma(a, yr0[qt][j], qa0[qt][u]);
              qa1[qt][u] = Tr<T>::mfma(a, yr1[qt][j], qa1[qt][u]);
            }
          }
          const bool wrap = s + RING >= nkc;
          const unsigned so = 64u * (unsigned)(wrap ? s + RING - nkc : s + RING);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            const unsigned vo = wrap ? voffn[qt] : voff[qt];
            yr0[qt][j] = srd_load16<V8>(y_srd, vo, so);
            yr1[qt][j] = srd_load16<V8>(y_srd, vo, row1 + so);
          }
        }
        // the next chunk's pieces are older than this chunk's 2 * CH * QT y refills: everything but those has landed
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * CH * QT) : "memory");
        __syncthreads();
      }
    }
    bool valid[QT];
    unsigned mbits[QT];
    V8 q0[QT][NKS], q1[QT][NKS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      valid[qt] = px_of(it, qt) < N;
      mbits[qt] = valid[qt] ? (mb[qt] & kmask) : 0u;
      voff[qt] = voffn[qt];
      voffn[qt] = voff_of(it + 2, qt);
      mb[qt] = mask_of(it + 1, qt);
#pragma unroll
      for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int t = 2 * s + (j >> 2);
          q0[qt][s][j] = (t < NDT) ? (T)qa0[qt][t][j & 3] : (T)0.0f;
          q1[qt][s][j] = (t < NDT) ? (T)qa1[qt][t][j & 3] : (T)0.0f;
        }
    }
    f32x4 au[QT][NDT], ac[QT][NDT];
    float w[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) w[qt] = 0.f;
    attend_staged<T, NDT, QT, 0>((const V8*)lds_ctx + lane, q0, kb4, sl2e, w, au, ac, sumrow);
    attend_staged<T, NDT, QT, 1>((const V8*)(lds_ctx + CB) + lane, q1, kb4, sl2e, w, au, ac, sumrow);
    for (int i = 0; i < K; ++i) {
      bool any = false;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) any = any || __ballot((mbits[qt] >> i) & 1u) != 0;
      if (!any) continue;                          // none of this wave's pixels inside disc i
      const float cw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), i));
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) w[qt] = ((mbits[qt] >> i) & 1u) ? cw : 0.f;
      gfr.soff = (unsigned)(((2 + i) * p.H + h) * CB);
      attend_staged<T, NDT, QT, 2>(gfr, q1, kb4, sl2e, w, au, ac, sumrow);
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
      if (valid[qt]) {
        T* obase = ob + (size_t)px_of(it, qt) * C + h * d;
        store_row16<T, NDT>(obase, au[qt], g, d);
        store_row16<T, NDT>(obase + (size_t)N * C, ac[qt], g, d);
      }
  }
}

template <typename T, int NDT, int NWV, int QT>
int launch_proj_wqs(PParams p, int n_img, hipStream_t st) {
  constexpr int TP = 16 * NWV * QT;
  p.tiles = (p.N + TP - 1) / TP;
  long wg_per_head = 256L / ((long)p.H * n_img);
  if (wg_per_head < 1) wg_per_head = 1;
  if (wg_per_head > p.tiles) wg_per_head = p.tiles;
  p.iters = (int)((p.tiles + wg_per_head - 1) / wg_per_head);
  if (const int v = g_sta_opt[STA_OPT_STAGED_TILES]) p.iters = v < p.tiles ? v : p.tiles;
  p.W = (p.tiles + p.iters - 1) / p.iters;
  const int lds = (2 * WQS_CH * NDT + 2 * fwd_frags(NDT)) * FRAG;
  static StaLdsAttr attr;
  if (!attr.ensure((const void*)xattn_fwd_proj_wqs_kernel<T, NDT, NWV, QT>, 160 * 1024)) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(fwd proj, streamed Wq) failed");
  hipLaunchKernelGGL((xattn_fwd_proj_wqs_kernel<T, NDT, NWV, QT>), dim3(p.W * p.H, n_img), dim3(64 * NWV), lds, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "fwd proj (streamed Wq) launch: %s", hipGetErrorString(e));
}

template <typename T, int NDT, int NWV, int RING, bool LL2 = false, bool YFRAG = false>
int launch_proj_cfg(PParams p, int n_img, int lds, hipStream_t st) {
  constexpr int TP = 16 * NWV;
  p.tiles = (p.N + TP - 1) / TP;
  // one workgroup per CU (its LDS image takes most of the 160 KiB): enough tiles per workgroup that ONE round of
  // workgroups covers the launch
  long wg_per_head = 256L / ((long)p.H * n_img);
  if (wg_per_head < 1) wg_per_head = 1;
  if (wg_per_head > p.tiles) wg_per_head = p.tiles;
  p.iters = (int)((p.tiles + wg_per_head - 1) / wg_per_head);
  if (const int v = g_sta_opt[STA_OPT_STAGED_TILES]) p.iters = v < p.tiles ? v : p.tiles;
  p.W = (p.tiles + p.iters - 1) / p.iters;
  static StaLdsAttr attr;
  if (!attr.ensure((const void*)xattn_fwd_proj_kernel<T, NDT, NWV, RING, LL2, YFRAG>, 160 * 1024))
    return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(fwd proj) failed");
  hipLaunchKernelGGL((xattn_fwd_proj_kernel<T, NDT, NWV, RING, LL2, YFRAG>), dim3(p.W * p.H, n_img), dim3(64 * NWV), lds, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "fwd proj launch: %s", hipGetErrorString(e));
}

template <typename T, int NDT>
int launch_proj(const PParams& p, int n_img, int lds, hipStream_t st) {
  // 8 waves (two per SIMD) x a 5-deep y ring ship; a 12-wave workgroup and a 10-deep ring measured no better in round 2
  // (profiles/r02_proj_fusion.md) and are gone. 4-wave workgroups stay reachable for small launches (tests).
  if (g_sta_opt[STA_OPT_STAGED_WAVES] == 4) return launch_proj_cfg<T, NDT, 4, 5>(p, n_img, lds, st);
  return launch_proj_cfg<T, NDT, 8, 5>(p, n_img, lds, st);
}

// The locals-from-L2 variant is built for the one shape that needs it (d = 80: SD-v1 level 1); 8 waves x a 5-deep y ring
template <typename T>
int launch_proj_ll2(const PParams& p, int n_img, int lds, hipStream_t st, bool qfrag) {
  if ((p.d + 15) / 16 != 5) return sta_fail(STA_E_UNSUP, "head dim %d: the locals-from-L2 projection-fused forward is built for 64 < d <= 80", p.d);
  return qfrag ? launch_proj_cfg<T, 5, 8, 5, true, true>(p, n_img, lds, st) : launch_proj_cfg<T, 5, 8, 5, true, false>(p, n_img, lds, st);
}

template <typename T>
int dispatch_proj(const PParams& p, int n_img, int lds, hipStream_t st) {
  switch ((p.d + 15) / 16) {
    case 1: return launch_proj<T, 1>(p, n_img, lds, st);
    case 2: return launch_proj<T, 2>(p, n_img, lds, st);
    case 3: return launch_proj<T, 3>(p, n_img, lds, st);
    case 4: return launch_proj<T, 4>(p, n_img, lds, st);
    case 5: return launch_proj<T, 5>(p, n_img, lds, st);
    case 6: return launch_proj<T, 6>(p, n_img, lds, st);
  }
  return sta_fail(STA_E_UNSUP, "head dim %d unsupported by the projection-fused forward", p.d);
}

// LDS plan of a launch: everything resident where that fits a CU (160 KiB); else, where the Wq slice and the two mandatory
// contexts fit (and the kernel exists: d = 80), those stay resident and the local contexts come from L2 (`ll2`).
constexpr int LDS_CU = 160 * 1024;
// does the shape take the streamed-Wq kernel? (Wq slice + the two mandatory contexts do not fit a CU: 144 < d <= 160 with C % 128 == 0)
bool proj_streams_wq(int C, int heads) {
  const int d = C / heads, ndt = (d + 15) / 16;
  return ndt == 10 && C % (32 * WQS_RING) == 0 && (2 * WQS_CH * ndt + 2 * fwd_frags(ndt)) * FRAG <= LDS_CU;
}
int proj_lds_bytes(int C, int heads, int K, bool* ll2 = nullptr) {
  const int d = C / heads, ndt = (d + 15) / 16;
  if (proj_streams_wq(C, heads)) {
    if (ll2) *ll2 = false;
    return (2 * WQS_CH * ndt + 2 * fwd_frags(ndt)) * FRAG;
  }
  const int full = (ndt * (C / 32) + (K + 2) * fwd_frags(ndt)) * FRAG;
  const int two = (ndt * (C / 32) + 2 * fwd_frags(ndt)) * FRAG;
  const bool l = full > LDS_CU && two <= LDS_CU && ndt == 5 && g_sta_opt[STA_OPT_PROJ_LL2] != 2;
  if (ll2) *ll2 = l;
  return l ? two : full;
}

int check_proj_shape(int N, int C, int heads, int M, int K) {
  if (N <= 0 || C <= 0 || heads <= 0 || M <= 0 || K < 0) return sta_fail(STA_E_ARG, "non-positive dimension");
  if (C % heads) return sta_fail(STA_E_ARG, "C=%d not divisible by heads=%d", C, heads);
  const int d = C / heads;
  if (d % 8 || (d > 96 && !proj_streams_wq(C, heads)))
    return sta_fail(STA_E_UNSUP, "head dim %d unsupported by the projection-fused forward (d %% 8 == 0; d <= 96, or 128 < d <= 160 with C %% 128 == 0: Wq streamed)", d);
  if (C % (32 * RING_MIN)) return sta_fail(STA_E_UNSUP, "C=%d unsupported by the projection-fused forward (need C %% %d == 0)", C, 32 * RING_MIN);
  if (M > STA_MAX_KEYS || M <= 16 * (NKT - 1)) return sta_fail(STA_E_UNSUP, "M=%d keys unsupported (65..%d)", M, STA_MAX_KEYS);
  if (K > STA_MAX_OBJECTS) return sta_fail(STA_E_UNSUP, "K=%d objects unsupported (max %d)", K, STA_MAX_OBJECTS);
  if (proj_lds_bytes(C, heads, K) > LDS_CU)
    return sta_fail(STA_E_UNSUP, "Wq slice + %d contexts need %d bytes of LDS (160 KiB per CU)", K + 2, proj_lds_bytes(C, heads, K));
  return STA_OK;
}

}  // namespace

extern "C" {

#ifdef STA_TRACE
// trace build only (tools/trace_proj.py): this translation unit's copy of the trace pointer
int sta_debug_set_trace_proj(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

int sta_xattn_fwd_proj_supported(int C, int heads, int M, int K) {
  const int rc = check_proj_shape(16, C, heads, M, K);
  g_sta_err[0] = 0;
  return rc == STA_OK;
}

int sta_xattn_fwd_proj_locals_from_l2(int C, int heads, int M, int K) {
  const int rc = check_proj_shape(16, C, heads, M, K);
  g_sta_err[0] = 0;
  bool ll2 = false;
  if (rc == STA_OK) proj_lds_bytes(C, heads, K, &ll2);
  return rc == STA_OK && ll2;
}

size_t sta_xattn_packed_wq_bytes(int C, int heads) {
  if (C <= 0 || heads <= 0 || C % heads || C % 32) return 0;
  const int d = C / heads;
  if (d % 8 || d > STA_MAX_HEAD_DIM) return 0;
  // + the head-PAIR fragments (sta_xattn_proj3.hip) where that kernel applies
  return (size_t)heads * ((d + 15) / 16) * (C / 32) * FRAG + (sta_p3::shape_ok(C, heads) ? sta_p3::wq_bytes(C, heads) : 0);
}

int sta_xattn_pack_wq(const void* wq, void* packed, int C, int heads, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!wq || !packed) return sta_fail(STA_E_ARG, "null pointer");
  if (sta_xattn_packed_wq_bytes(C, heads) == 0) return sta_fail(STA_E_UNSUP, "C=%d heads=%d unsupported", C, heads);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const int d = C / heads, ndt = (d + 15) / 16;
  const dim3 grid(ndt * (C / 32), heads);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(pack_wq_kernel<__bf16>, grid, dim3(64), 0, st, (const __bf16*)wq, (__bf16*)packed, C, d, ndt);
  else
    hipLaunchKernelGGL(pack_wq_kernel<_Float16>, grid, dim3(64), 0, st, (const _Float16*)wq, (_Float16*)packed, C, d, ndt);
  if (sta_p3::shape_ok(C, heads)) {      // the same weights per head PAIR: 2d = 80 output columns = 5 tiles, no padding
    char* pair = (char*)packed + (size_t)heads * ndt * (C / 32) * FRAG;
    const dim3 g2(sta_p3::NT * (C / 32), heads / 2);
    if (dtype == STA_BF16)
      hipLaunchKernelGGL(pack_wq_kernel<__bf16>, g2, dim3(64), 0, st, (const __bf16*)wq, (__bf16*)pair, C, 2 * d, sta_p3::NT);
    else
      hipLaunchKernelGGL(pack_wq_kernel<_Float16>, g2, dim3(64), 0, st, (const _Float16*)wq, (_Float16*)pair, C, 2 * d, sta_p3::NT);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "pack_wq launch: %s", hipGetErrorString(e));
}

size_t sta_xattn_packed_kv_proj_bytes(int n_ctx, int heads, int d) {
  if (n_ctx <= 0 || heads <= 0 || d <= 0 || d % 8 || d > STA_MAX_HEAD_DIM) return 0;
  // + the per-(ctx, head) blocks of the head-pair kernel where it applies (d = 40, even head count)
  return (size_t)n_ctx * heads * fwd_frags((d + 15) / 16) * FRAG + ((d == sta_p3::D && heads % 2 == 0) ? sta_p3::kv_bytes(n_ctx, heads) : 0);
}

int sta_xattn_pack_kv_proj(const void* k, const void* v, void* packed, int n_ctx, int M, int C, int heads, int dtype,
                           void* stream) {
  g_sta_err[0] = 0;
  if (!k || !v || !packed) return sta_fail(STA_E_ARG, "null pointer");
  if (n_ctx <= 0) return sta_fail(STA_E_ARG, "n_ctx=%d", n_ctx);
  if (C <= 0 || heads <= 0 || C % heads) return sta_fail(STA_E_ARG, "C=%d heads=%d", C, heads);
  const int d = C / heads, ndt = (d + 15) / 16;
  if (sta_xattn_packed_kv_proj_bytes(n_ctx, heads, d) == 0) return sta_fail(STA_E_UNSUP, "head dim %d unsupported", d);
  if (M <= 0 || M > STA_MAX_KEYS) return sta_fail(STA_E_UNSUP, "M=%d keys unsupported (max %d)", M, STA_MAX_KEYS);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const dim3 grid(fwd_frags(ndt), n_ctx * heads);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(pack_kv_proj_kernel<__bf16>, grid, dim3(64), 0, st, (const __bf16*)k, (const __bf16*)v,
                       (__bf16*)packed, M, C, heads, d, ndt);
  else
    hipLaunchKernelGGL(pack_kv_proj_kernel<_Float16>, grid, dim3(64), 0, st, (const _Float16*)k, (const _Float16*)v,
                       (_Float16*)packed, M, C, heads, d, ndt);
  if (d == sta_p3::D && heads % 2 == 0 && M <= sta_p3::KR) {
    if (int rc = sta_p3::pack_kv(k, v, (char*)packed + (size_t)n_ctx * heads * fwd_frags(ndt) * FRAG, n_ctx, M, C, heads, dtype, st)) return rc;
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "pack_kv_proj launch: %s", hipGetErrorString(e));
}

// does a launch of this shape take the head-pair kernel (sta_xattn_proj3.hip)?
static bool takes_pair_kernel(int n_img, int N, int C, int heads, int M, int K) {
  const long pair_wgs = (long)((N + 127) / 128) * (heads / 2) * n_img;
  return sta_p3::eligible(C, heads, M, K) && g_sta_opt[STA_OPT_PROJ_PAIR] != 2 && (pair_wgs >= 256 || g_sta_opt[STA_OPT_PROJ_PAIR] == 1);
}

static int fwd_proj_impl(const void* y, const void* packed_wq, const void* packed_kv, const uint8_t* mask,
                         const float* coef, void* out, int n_img, int N, int C, int heads, int M, int K, float scale,
                         int dtype, void* stream, bool qfrag, bool ofrag = false, void* stats = nullptr);

int sta_xattn_fwd_proj_qfrag_supported(int n_img, int N, int C, int heads, int M, int K) {
  const int rc = n_img < 1 ? STA_E_ARG : check_proj_shape(N, C, heads, M, K);
  g_sta_err[0] = 0;
  bool ll2 = false;
  if (rc == STA_OK) proj_lds_bytes(C, heads, K, &ll2);
  return rc == STA_OK && N % 16 == 0 && (ll2 || takes_pair_kernel(n_img, N, C, heads, M, K));
}

int sta_xattn_fwd_proj_qfrag(const void* y, const void* packed_wq, const void* packed_kv, const uint8_t* mask,
                             const float* coef, void* out, int n_img, int N, int C, int heads, int M, int K, float scale,
                             int dtype, void* stream) {
  return fwd_proj_impl(y, packed_wq, packed_kv, mask, coef, out, n_img, N, C, heads, M, K, scale, dtype, stream, true);
}

int sta_xattn_fwd_proj_qfrag_ofrag(const void* y, const void* packed_wq, const void* packed_kv, const uint8_t* mask,
                                   const float* coef, void* out, int n_img, int N, int C, int heads, int M, int K, float scale,
                                   int dtype, void* stream) {
  return fwd_proj_impl(y, packed_wq, packed_kv, mask, coef, out, n_img, N, C, heads, M, K, scale, dtype, stream, true, true);
}

int sta_xattn_fwd_proj_ex(const void* y, const void* packed_wq, const void* packed_kv, const uint8_t* mask,
                          const float* coef, void* out, int n_img, int N, int C, int heads, int M, int K, float scale,
                          int dtype, int layout, void* stats, void* stream) {
  if (layout < 0 || layout > 2) return sta_fail(STA_E_ARG, "layout %d (0 row-major, 1 query fragments in, 2 + out fragments out)", layout);
  return fwd_proj_impl(y, packed_wq, packed_kv, mask, coef, out, n_img, N, C, heads, M, K, scale, dtype, stream, layout >= 1, layout == 2, stats);
}

int sta_xattn_fwd_proj(const void* y, const void* packed_wq, const void* packed_kv, const uint8_t* mask,
                       const float* coef, void* out, int n_img, int N, int C, int heads, int M, int K, float scale,
                       int dtype, void* stream) {
  return fwd_proj_impl(y, packed_wq, packed_kv, mask, coef, out, n_img, N, C, heads, M, K, scale, dtype, stream, false);
}

}  // extern "C"

static int fwd_proj_impl(const void* y, const void* packed_wq, const void* packed_kv, const uint8_t* mask,
                         const float* coef, void* out, int n_img, int N, int C, int heads, int M, int K, float scale,
                         int dtype, void* stream, bool qfrag, bool ofrag, void* stats) {
  g_sta_err[0] = 0;
  if (!y || !packed_wq || !packed_kv || !out) return sta_fail(STA_E_ARG, "null pointer");
  if (n_img < 1 || n_img > 65535) return sta_fail(STA_E_ARG, "n_img=%d", n_img);
  if (int rc = check_proj_shape(N, C, heads, M, K)) return rc;
  if (K > 0 && (!mask || !coef)) return sta_fail(STA_E_ARG, "mask/coef required when K > 0");
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  if ((size_t)2 * N * C * 2 >= 0xfffffff0ull) return sta_fail(STA_E_UNSUP, "one image of activations must stay below 4 GiB");
  PParams p{};
  p.y = y; p.wq = (const char*)packed_wq; p.kv = (const char*)packed_kv; p.mask = mask; p.coef = coef; p.out = out;
  if (K == 0) {  // unconditional prologue loads: readable (ignored) bytes
    p.mask = (const uint8_t*)y;
    p.coef = (const float*)y;
  }
  p.N = N; p.C = C; p.H = heads; p.d = C / heads; p.M = M; p.K = K; p.nkc = C / 32;
  p.sl2e = scale * 1.4426950408889634f;
  bool ll2 = false;
  const int lds = proj_lds_bytes(C, heads, K, &ll2);
  hipStream_t st = (hipStream_t)stream;
  // Head pairs share one read of y where both heads' operands fit a CU (d = 40, C = 160 / 320, K <= 2: sta_xattn_proj3.hip)
  // and the launch still fills the chip with one pair workgroup per CU (level 0 at 32 / 16 images: 147 / 72 us against
  // 156 / 92 us one head per workgroup; a one-image launch has 128 pair workgroups and takes the one-head kernel).
  if (takes_pair_kernel(n_img, N, C, heads, M, K)) {
    const int ndt = (p.d + 15) / 16;
    const char* wq_pair = (const char*)packed_wq + (size_t)heads * ndt * (C / 32) * FRAG;
    const char* kv3 = (const char*)packed_kv + (size_t)n_img * (K + 2) * heads * fwd_frags(ndt) * FRAG;
    return sta_p3::forward(y, wq_pair, kv3, p.mask, p.coef, out, n_img, N, C, heads, M, K, p.sl2e, dtype, st, qfrag, ofrag, (unsigned*)stats);
  }
  if (proj_streams_wq(C, heads)) {      // SD-v1 levels 2 / mid: Wq streamed through an LDS ring, two contexts resident, locals from L2
    if (qfrag || ofrag) return sta_fail(STA_E_UNSUP, "the streamed-Wq kernel reads y row-major");
    // STA_OPT_STAGED_QT = 2: four waves x two 16-pixel tiles per wave (one wave per SIMD, each Wq fragment read serves four MFMAs);
    // default: eight waves x one tile
    if (g_sta_opt[STA_OPT_STAGED_QT] == 2)
      return dtype == STA_BF16 ? launch_proj_wqs<__bf16, 10, 4, 2>(p, n_img, st) : launch_proj_wqs<_Float16, 10, 4, 2>(p, n_img, st);
    return dtype == STA_BF16 ? launch_proj_wqs<__bf16, 10, 8, 1>(p, n_img, st) : launch_proj_wqs<_Float16, 10, 8, 1>(p, n_img, st);
  }
  if (ll2) {      // SD-v1 level 1: Wq + the two mandatory contexts resident, local contexts from L2
    if (qfrag && N % 16) return sta_fail(STA_E_UNSUP, "query-fragment order needs N %% 16 == 0 (N=%d)", N);
    if (ofrag) return sta_fail(STA_E_UNSUP, "out-fragment order is written by the head-pair kernel only");
    return dtype == STA_BF16 ? launch_proj_ll2<__bf16>(p, n_img, lds, st, qfrag) : launch_proj_ll2<_Float16>(p, n_img, lds, st, qfrag);
  }
  if (qfrag) return sta_fail(STA_E_UNSUP, "query-fragment order is read by the head-pair and the locals-from-L2 kernels only (sta_xattn_fwd_proj_qfrag_supported)");
  return dtype == STA_BF16 ? dispatch_proj<__bf16>(p, n_img, lds, st) : dispatch_proj<_Float16>(p, n_img, lds, st);
}
