// sta_conv.hip — the 3x3, stride-1, padding-1 convolutions of the UNet's ResBlocks / Upsample layers on NHWC activations as an
// implicit GEMM on gfx950 MFMAs (reference: ResBlock in_layers / out_layers and Upsample.conv, openaimodel.py:163-275, :91-120;
// called once per ResBlock half inside UNetModel.forward :710-743). These convolutions are ~45 % of a UNet call's GPU time through
// the library kernels (17 - 27 % of the MFMA peak at the UNet's shapes); this is the same operation laid out for this chip.
//
//   out[b][y][x][o] = sum_{ky, kx, i} w[o][i][ky][kx] * in[b][y + ky - 1][x + kx - 1][i]          (zero outside the image)
//
// Pixels are the MFMA columns: Out^T[o][px] = sum over (tap, 32-channel step) of W_tap[o][i] . X_tap^T[i][px]  (v_mfma_f32_16x16x32).
//   * A workgroup (8 waves) owns an output tile of 256 pixels (8 rows x 32 columns, 16 x 16 for 16-pixel-wide images, or two whole
//     8 x 8 images = 128 pixels) and 160 output channels (128 for the VAE decoder's 128 / 256 / 512); a wave owns 64 of the pixels
//     (four 16-pixel row segments) and 80 (64) of the channels: 20 (16) accumulator tiles.
//   * The input tile WITH its one-pixel halo ((8+2) x (32+2) pixels x 32 channels = 21.25 KiB per channel step) is copied to LDS
//     once per channel step by LDS-DMA with per-lane source addresses (pixels outside the image read a page of zeros), double
//     buffered; all nine taps read their B operands from it: one ds_read_b128 per (tap, pixel segment), with the four 16-byte
//     channel chunks of a pixel stored at slot g ^ 2 ((p >> 2) & 1) so that the read is bank-conflict-free at every tap offset.
//   * The weights are re-laid out once per model into 1-KiB A-operand fragments [part][channel step][ky][kx][tile] and streamed
//     through a 2-slot LDS ring, one kernel row (3 taps x 10 tiles = 30 KiB) per step; a step is 60 MFMAs per wave behind
//     27 operand reads (those of tap kx + 1 requested before the MFMAs of tap kx), one barrier per step, the next step's DMA issued
//     behind the first tap's MFMAs (at the barrier every wave would pay its issue cost with the matrix pipe idle).
//   * Workgroups are persistent; the (tile, part) -> workgroup map keeps the parts of one pixel tile on one XCD (shared L2).
//   * `up2`: the input is the nearest-neighbour 2x upsampling of a half-resolution tensor (Upsample.forward, :107-120): the
//     halo copy reads pixel (y >> 1, x >> 1) of the small tensor, so the upsampled tensor never exists in HBM.
//   * Epilogue: + bias + residual tensor (ResBlock's `skip_connection(x) + out_layers(h)`), and — `stats` — the partial sums / sums of
//     squares per output channel of the values it stores: the statistics of the GroupNorm that consumes the result.
//   * The input gradient of this convolution is the same convolution with the weight's channel axes exchanged and its taps mirrored:
//     the tracked epochs run forward and backward on this kernel (sta.fused.Conv3x3Fn).
//
// Roofline: MFMA (2 * 9 * Cin * Cout flop per pixel against 2 (Cin + Cout) bytes: 1440 flop/B at 320 -> 320).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "sta_xattn.h"
#include "sta_unet.h"
#include "sta_internal.h"
#include "sta_xattn_dev.h"

namespace {

constexpr int CV_NW = 8;
// NTW = row tiles (16 output channels) per wave: 5 (a workgroup owns 160 channels: the UNet's 320 / 640 / 1280) or 4 (128 channels:
// the VAE decoder's 128 / 256 / 512)
constexpr int cv_part(int ntw) { return 32 * ntw; }                       // output channels per workgroup
constexpr int cv_nt(int ntw) { return 2 * ntw; }                          // row tiles per part
constexpr int cv_wfr(int ntw) { return 3 * cv_nt(ntw); }                  // 30 / 24 weight fragments per step (one kernel row)
constexpr int cv_wper(int ntw) { return (cv_wfr(ntw) + CV_NW - 1) / CV_NW; }   // 4 / 3 weight DMAs per wave per step
constexpr int cv_wslot(int ntw) { return cv_wper(ntw) * CV_NW * FRAG; }   // 32 / 24 KiB
// Tile geometries. GEO 0: 8 rows x 32 columns; GEO 1: 16 x 16 (16-pixel-wide images): 256 pixels, four 16-pixel segments per wave.
// GEO 2: TWO whole 8 x 8 images (the UNet's lowest level): 128 pixels, two segments (four image rows) per wave — at 64 images x 1280
// channels that is 256 work items, one per CU, where 256-pixel tiles would leave half of the chip idle. The halo tile is stored at a
// row pitch that keeps the chunk swizzle separable (see seg_e below): 36 / 20 / 16 pixels. XPW = input DMA pieces (16 pixels each)
// per wave per channel step; NQ = pixel segments per wave.
constexpr int cv_tr(int geo) { return geo == 0 ? 8 : geo == 1 ? 16 : 8; }
constexpr int cv_tc(int geo) { return geo == 0 ? 32 : geo == 1 ? 16 : 8; }
constexpr int cv_pitch(int geo) { return geo == 0 ? 36 : geo == 1 ? 20 : 16; }
constexpr int cv_npx(int geo) { return geo == 2 ? 2 * 10 * 16 : (cv_tr(geo) + 2) * cv_pitch(geo); }    // 360 / 360 / 320 pixels
constexpr int cv_nq(int geo) { return geo == 2 ? 2 : 4; }
constexpr int cv_xpw(int geo) { return (cv_npx(geo) + 16 * CV_NW - 1) / (16 * CV_NW); }                   // 3
constexpr int cv_xbuf(int geo) { return cv_xpw(geo) * CV_NW * FRAG; }                                      // 24 KiB
constexpr int cv_lds(int ntw, int geo) { return 2 * cv_wslot(ntw) + 2 * cv_xbuf(geo); }                    // 112 / 96 KiB
#ifndef CV_STAGE_AFTER_TAP
#define CV_STAGE_AFTER_TAP 0                  // the next step's DMA is issued behind the MFMAs of this tap (not right behind the barrier,
#endif                                        // where every wave of the workgroup would pay the issue cost with the matrix pipe idle)

// conv weight (o, i, ky, kx) at o*so + i*si + ky*sy + kx*sx -> fragments [part][kc][ky][kx][tile t]: lane (g, c) holds
// W[16 nt part + 16 t + c][32 kc + 8 g .. + 7][ky][kx]      (nt = 10 or 8 row tiles per part)
template <typename T>
__global__ __launch_bounds__(64) void pack_conv_w_kernel(const T* __restrict__ w, long so, long si, long sy, long sx, T* __restrict__ packed,
                                                         int nkc, int nt) {
  const int fr = blockIdx.x;                   // (((part * nkc + kc) * 3 + ky) * 3 + kx) * nt + t
  const int t = fr % nt, kx = (fr / nt) % 3, ky = (fr / (3 * nt)) % 3, kc = (fr / (9 * nt)) % nkc, part = fr / (9 * nt * nkc);
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const T* src = w + (size_t)(16 * nt * part + 16 * t + c) * so + (size_t)(32 * kc + 8 * g) * si + ky * sy + kx * sx;
  typename Tr<T>::V8 x;
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = src[j * si];
  *(typename Tr<T>::V8*)(packed + (size_t)fr * (FRAG / 2) + lane * 8) = x;
}

// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), result in every lane: xor 1, xor 2, half-row mirror, row mirror
__device__ __forceinline__ float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));   // row_half_mirror
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));   // row_mirror
  return x;
}

struct CV {
  const char* x;        // [B][Hs][Ws][Cin] (Hs = H >> up2)
  const char* w;        // packed weights
  const char* zeros;    // >= 2 * Cin bytes of zeros
  void* out;            // [B][H][W][Cout]
  const void* bias;     // [Cout] or null
  const void* res;      // [B][H][W][Cout] or null: out = conv + bias + res
  float* stats;         // null, or [B + 1][stats_slots][Cout][2] fp32: per (image, slot) partial sums / sums of squares of the STORED values
  int stats_slots;
  int B, H, W, Cin, Cout, up2;
  int parts, tiles_x, tiles_per_img, items, xcd_map;
};

template <typename T, int GEO, int NTW>
__global__ __launch_bounds__(64 * CV_NW, 1) void conv3x3_nhwc_kernel(const CV p) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  constexpr int CV_PART = cv_part(NTW), CV_NT = cv_nt(NTW), CV_WFR = cv_wfr(NTW), CV_WPER = cv_wper(NTW), CV_WSLOT = cv_wslot(NTW);
  // halo tile: (TR + 2) rows of TC + 2 pixels at a pitch that is a multiple of 4: the chunk swizzle ((pixel index >> 2) & 1) << 1 of pixel
  // (row, xx) is then ((row * (PITCH / 4) + (xx >> 2)) & 1) << 1 — separable in row and column, so a lane needs 12 operand addresses
  // instead of 36 (PITCH / 4 odd: bit 5 of the address flips on odd kernel rows; even: it does not depend on the row at all)
  constexpr int TR = cv_tr(GEO), TC = cv_tc(GEO), TWH = TC + 2, PITCH = cv_pitch(GEO), NPX = cv_npx(GEO), XPW = cv_xpw(GEO), CV_XBUF = cv_xbuf(GEO), NQ = cv_nq(GEO);
  constexpr bool ROW_FLIPS = (PITCH / 4) % 2 == 1;
  static_assert(PITCH % 4 == 0 && PITCH >= TWH, "halo tile geometry");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int pq = wv & 3, ch = wv >> 2;                     // pixel quarter (4 row segments), channel half (5 row tiles)
  char* xring = smem;                                      // input buffers first: their operand reads then fit ds_read's 16-bit offset field
  char* wring = smem + 2 * CV_XBUF;
  const int nkc = p.Cin >> 5;
  const int Hs = p.H >> p.up2, Ws = p.W >> p.up2;
  const __amdgpu_buffer_rsrc_t w_srd = make_srd(p.w, (unsigned)((size_t)p.parts * nkc * 9 * CV_NT * FRAG));
  const unsigned lane16 = (unsigned)lane * 16u;

  // item -> (tile, part): the parts of a pixel tile run back to back on ONE XCD (workgroup id % 8), so the input tile is fetched
  // from HBM once per XCD-L2 and the weight stream of a part is shared by the XCD's CUs walking the channel steps together
  // (tile counts that are not a multiple of 8 — small batches — take the plain order)
  auto item_tile = [&](int it, int& tile, int& part) {
    if (p.xcd_map) {
      const int j = it >> 3;
      tile = (j / p.parts) * 8 + (it & 7);
      part = j % p.parts;
    } else {
      tile = it / p.parts;
      part = it - tile * p.parts;
    }
  };
  // per-lane source pointer of input piece `pc` (pixels 16 pc .. + 15 of the halo tile, this lane: pixel 16 pc + (lane >> 2), LDS slot
  // lane & 3) at channel step 0
  auto in_ptr = [&](int tile, int pc) -> const char* {
    const int pp = 16 * pc + (lane >> 2);
    const int chunk = (lane & 3) ^ (((pp >> 2) & 1) << 1);
    int b, y, x, hx;
    if (GEO == 2) {                                        // image 2 tile + pp / 160, halo row (pp % 160) / 16
      const int im = pp / (10 * PITCH), rem = pp - im * (10 * PITCH);
      const int hy = rem / PITCH;
      hx = rem - hy * PITCH;
      b = 2 * tile + im; y = hy - 1; x = hx - 1;
    } else {
      b = tile / p.tiles_per_img;
      const int tt = tile - b * p.tiles_per_img;
      const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
      const int hy = pp / PITCH;
      hx = pp - hy * PITCH;
      y = ty * TR - 1 + hy; x = tx * TC - 1 + hx;
    }
    const bool ok = pp < NPX && hx < TWH && y >= 0 && y < p.H && x >= 0 && x < p.W && b < p.B;
    const size_t px = ((size_t)b * Hs + (y >> p.up2)) * Ws + (x >> p.up2);
    return ok ? p.x + px * (size_t)p.Cin * sizeof(T) + chunk * 16 : p.zeros + chunk * 16;
  };
  auto stage_w = [&](int part, int kc, int ky, int slot) __attribute__((always_inline)) {
    const unsigned base = (unsigned)(((part * nkc + kc) * 3 + ky) * CV_WFR) * (unsigned)FRAG;
#pragma unroll
    for (int i = 0; i < CV_WPER; ++i) {
      const int f = wv + CV_NW * i;
      const int fs = f < CV_WFR ? f : 0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (__attribute__((address_space(3))) void*)(wring + slot * CV_WSLOT + f * FRAG), 16, lane16,
                                               base + (unsigned)fs * (unsigned)FRAG, 0, 0);
    }
  };
  auto stage_x = [&](const char* src, int pc, int buf) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(xring + buf * CV_XBUF + pc * FRAG), 16, 0, 0);
  };

  // this wave's NQ pixel segments: segment index gi = 4 pq + q -> (row, first column) inside the tile (GEO 2: image pq >> 1, rows
  // 4 (pq & 1) + 2 q and the next, lane c -> (row + (c >> 3), c & 7)). Byte offset inside an input buffer of this lane's B operand chunk for tap (ky, kx):
  // seg_e[q][kx] + ky * PITCH * 64, bit 5 flipped when ky is odd and ROW_FLIPS
  auto seg_rc = [&](int q, int& ry, int& x0) {             // first halo row / column of the segment's lane-0 pixel at tap (0, 0)
    const int gi = 4 * pq + q;
    if (GEO == 0) { ry = gi >> 1; x0 = (gi & 1) * 16 + c16; }
    else if (GEO == 1) { ry = gi; x0 = c16; }
    else { ry = 10 * (pq >> 1) + 4 * (pq & 1) + 2 * q + (c16 >> 3); x0 = c16 & 7; }
  };
  unsigned seg_e[NQ][3];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int ry, x0;
    seg_rc(q, ry, x0);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xx = x0 + kx;
      seg_e[q][kx] = (unsigned)((ry * PITCH + xx) * 64 + ((g ^ (((ry * (PITCH / 4) + (xx >> 2)) & 1) << 1)) << 4));
    }
  }

  int it = blockIdx.x;
  if (it >= p.items) return;
  int tile, part;
  item_tile(it, tile, part);
  const char* xp[XPW];                                     // this wave's input pieces (wv, wv + 8, ...) of the NEXT channel step
#pragma unroll
  for (int i = 0; i < XPW; ++i) xp[i] = in_ptr(tile, wv + CV_NW * i);
  // prologue: kernel row 0 of channel step 0 and the whole input tile of channel step 0
  stage_w(part, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < XPW; ++i) { stage_x(xp[i], wv + CV_NW * i, 0); xp[i] += 64; }

  const size_t obytes = (size_t)p.B * p.H * p.W * p.Cout * sizeof(T);
  const __amdgpu_buffer_rsrc_t o_srd = make_srd(p.out, (unsigned)(obytes < 0xfffffff0ull ? obytes : 0xfffffff0ull));
  const __amdgpu_buffer_rsrc_t r_srd = make_srd(p.res ? p.res : p.out, (unsigned)(obytes < 0xfffffff0ull ? obytes : 0xfffffff0ull));
  bool first_of_tile = false;                              // the step that follows an epilogue: 4 NTW stores are younger than its DMAs

  while (true) {
    f32x4 acc[NQ][NTW];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    int nit = it + gridDim.x, ntile = 0, npart = 0;
    const bool more = nit < p.items;
    if (more) item_tile(nit, ntile, npart);

    // one step = one kernel row (3 taps) of one channel step. Slots are compile-time: two channel steps (6 steps) per trip.
    auto step = [&](auto xb_tag, auto ws_tag, auto ky_tag, const int kc) __attribute__((always_inline)) {
      constexpr int XB = decltype(xb_tag)::value, WS = decltype(ws_tag)::value, KY = decltype(ky_tag)::value;
      if (first_of_tile) {                                 // the epilogue's stores (and statistics atomics) are younger than this step's DMAs
        if (p.stats) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ * NTW + 2 * NTW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ * NTW) : "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      first_of_tile = false;
      __builtin_amdgcn_s_barrier();
      // the next step's kernel row, and one piece of the next channel step's input tile
      auto stage_next = [&]() __attribute__((always_inline)) {
        const bool last_kc = kc + 1 == nkc;
        if (KY < 2) stage_w(part, kc, KY + 1, WS ^ 1);
        else if (!last_kc) stage_w(part, kc + 1, 0, WS ^ 1);
        else if (more) stage_w(npart, 0, 0, WS ^ 1);
        if (!last_kc || more) {
#pragma unroll
          for (int i = KY; i < XPW; i += 3) {              // pieces KY, KY + 3 of this wave: one or two per step
            if (last_kc) xp[i] = in_ptr(ntile, wv + CV_NW * i);
            stage_x(xp[i], wv + CV_NW * i, XB ^ 1);
            xp[i] += 64;
          }
        }
      };
      if (CV_STAGE_AFTER_TAP < 0) stage_next();
      const char* xb = xring + XB * CV_XBUF;
      const V8* wf = (const V8*)(wring + WS * CV_WSLOT + lane * 16) + (NTW * ch) * 64;
      // operands of tap kx + 1 are requested before the 20 MFMAs of tap kx (two register sets)
      V8 a[NTW], b[NQ];
      auto load_tap = [&](int kx, V8 (&aa)[NTW], V8 (&bb)[NQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) bb[q] = *(const V8*)(xb + KY * PITCH * 64 + ((KY & 1) && ROW_FLIPS ? seg_e[q][kx] ^ 32u : seg_e[q][kx]));
#pragma unroll
        for (int t = 0; t < NTW; ++t) aa[t] = wf[(kx * CV_NT + t) * 64];
      };
      load_tap(0, a, b);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        V8 an[NTW], bn[NQ];
        if (kx < 2) load_tap(kx + 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
          for (int q = 0; q < NQ; ++q) acc[q][t] = Tr<T>::mfma(a[t], b[q], acc[q][t]);
        __builtin_amdgcn_sched_barrier(0);
        if (kx == CV_STAGE_AFTER_TAP) {
          stage_next();
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kx < 2) {
#pragma unroll
          for (int t = 0; t < NTW; ++t) a[t] = an[t];
#pragma unroll
          for (int q = 0; q < NQ; ++q) b[q] = bn[q];
        }
      }
    };
    for (int kc = 0; kc < nkc; kc += 2) {
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      step(I0{}, I0{}, I0{}, kc);
      step(I0{}, I1{}, I1{}, kc);
      step(I0{}, I0{}, I2{}, kc);
      step(I1{}, I1{}, I0{}, kc + 1);
      step(I1{}, I0{}, I1{}, kc + 1);
      step(I1{}, I1{}, I2{}, kc + 1);
    }
    // epilogue: lane (g, c) of tile t holds output channels 16 t + 4 g .. + 3 of pixel c (+ bias, + the residual tensor):
    // ALWAYS NQ * NTW stores of 8 bytes, issued behind every load of the epilogue
    {
      const int b = GEO == 2 ? 2 * tile + (pq >> 1) : tile / p.tiles_per_img, tt = GEO == 2 ? 0 : tile - b * p.tiles_per_img;
      const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      float bs[NTW][4];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        V4 bv = {};
        if (p.bias) bv = *(const V4*)((const T*)p.bias + part * CV_PART + ch * (16 * NTW) + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[t][r] = (float)bv[r];
      }
      float ssum[NTW][4], ssq[NTW][4];                     // GroupNorm statistics of the stored values, per output channel
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[t][r] = 0.f; ssq[t][r] = 0.f; }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        int ry, x0;
        seg_rc(q, ry, x0);
        if (GEO == 2) ry -= 10 * (pq >> 1);
        const size_t px = ((size_t)b * p.H + ty * TR + ry) * p.W + tx * TC + x0;
        const bool img_ok = b < p.B;                       // (GEO 2: the second image of the last tile of an odd batch does not exist)
        const unsigned base = (unsigned)((px * p.Cout + part * CV_PART + ch * (16 * NTW) + 4 * g) * sizeof(T));
        V4 rv[NTW];
        if (p.res) {
#pragma unroll
          for (int t = 0; t < NTW; ++t)
            rv[t] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b64(r_srd, img_ok ? base + (unsigned)(16 * t * sizeof(T)) : 0xfffffff0u, 0, 0));
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          V4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (T)(acc[q][t][r] + bs[t][r] + (p.res ? (float)rv[t][r] : 0.f));
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), o_srd, img_ok ? base + (unsigned)(16 * t * sizeof(T)) : 0xfffffff0u, 0, 0);
          if (p.stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = img_ok ? (float)o[r] : 0.f;
              ssum[t][r] += v;
              ssq[t][r] += v * v;
            }
          }
        }
      }
      // the consumer's GroupNorm statistics from here (sta_stats_finalize -> sta_groupnorm_silu_nhwc_cstats): sum over the wave's 16 pixel
      // lanes, then lane c16 == 0 stores the wave's partial sums of its 4 x NTW channels into the slot (tile of the image, pixel quarter)
      // — plain stores, no atomics (2.6 M atomics per level-0 convolution cost 18 % of it), ALWAYS 2 NTW store instructions per wave
      if (p.stats) {
        const int slot = GEO == 2 ? (pq & 1) : tt * 4 + pq;
        const int bb = b < p.B ? b : p.B;                  // a nonexistent image (GEO 2, odd batch) writes the spare image slot
        float* dst = p.stats + (((size_t)bb * p.stats_slots + slot) * p.Cout + part * CV_PART + ch * (16 * NTW) + 4 * g) * 2;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          f32x4 lo, hi;
          lo[0] = row16_sum(ssum[t][0]); lo[1] = row16_sum(ssq[t][0]); lo[2] = row16_sum(ssum[t][1]); lo[3] = row16_sum(ssq[t][1]);
          hi[0] = row16_sum(ssum[t][2]); hi[1] = row16_sum(ssq[t][2]); hi[2] = row16_sum(ssum[t][3]); hi[3] = row16_sum(ssq[t][3]);
          if (c16 == 0) {
            *(f32x4*)(dst + 32 * t) = lo;
            *(f32x4*)(dst + 32 * t + 4) = hi;
          }
        }
      }
    }
    if (!more) break;
    it = nit; tile = ntile; part = npart;
    first_of_tile = true;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int conv_geo(int H, int W) {                  // tile geometry of an H x W image, -1: none
  if (W % 32 == 0 && H % 8 == 0) return 0;
  if (W == 16 && H % 16 == 0) return 1;
  if (W == 8 && H == 8) return 2;
  return -1;
}

}  // namespace

extern "C" {

static int conv_ntw(int Cout) { return Cout > 0 && Cout % 160 == 0 ? 5 : (Cout > 0 && Cout % 128 == 0 ? 4 : 0); }

int sta_conv3x3_nhwc_supported(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || conv_geo(H, W) < 0) return 0;
  if (Cin <= 0 || Cin % 64 || conv_ntw(Cout) == 0) return 0;
  if ((size_t)B * H * W * (size_t)Cout * 2 >= 0xfffffff0ull) return 0;
  return 1;
}

int sta_conv3x3_stats_slots(int H, int W) {
  const int geo = conv_geo(H, W);
  return geo < 0 ? 0 : geo == 2 ? 2 : 4 * (H / cv_tr(geo)) * (W / cv_tc(geo));
}

size_t sta_conv3x3_packed_w_bytes(int Cin, int Cout) {
  return (Cin > 0 && Cin % 64 == 0 && conv_ntw(Cout)) ? (size_t)Cout * Cin * 9 * 2 : 0;
}

int sta_conv3x3_pack_w(const void* w, long so, long si, long sy, long sx, void* packed, int Cin, int Cout, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!w || !packed) return sta_fail(STA_E_ARG, "null pointer");
  if (sta_conv3x3_packed_w_bytes(Cin, Cout) == 0)
    return sta_fail(STA_E_UNSUP, "conv3x3: Cin %% 64 == 0 and Cout %% 160 == 0 or Cout %% 128 == 0 (Cin=%d Cout=%d)", Cin, Cout);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int nkc = Cin / 32, nt = cv_nt(conv_ntw(Cout));
  const unsigned nfr = (unsigned)(Cout / (16 * nt)) * nkc * 9 * nt;
  if (dtype == STA_BF16) hipLaunchKernelGGL(pack_conv_w_kernel<__bf16>, dim3(nfr), dim3(64), 0, st, (const __bf16*)w, so, si, sy, sx, (__bf16*)packed, nkc, nt);
  else hipLaunchKernelGGL(pack_conv_w_kernel<_Float16>, dim3(nfr), dim3(64), 0, st, (const _Float16*)w, so, si, sy, sx, (_Float16*)packed, nkc, nt);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "pack_conv_w launch: %s", hipGetErrorString(e));
}

int sta_conv3x3_nhwc(const void* x, const void* packed_w, const void* zeros, const void* bias, const void* res, void* out, float* stats, int B,
                     int H, int W, int Cin, int Cout, int up2, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!x || !packed_w || !zeros || !out) return sta_fail(STA_E_ARG, "null pointer");
  if (!sta_conv3x3_nhwc_supported(B, H, W, Cin, Cout))
    return sta_fail(STA_E_UNSUP, "conv3x3_nhwc: unsupported geometry B=%d H=%d W=%d Cin=%d Cout=%d", B, H, W, Cin, Cout);
  if (up2 && (H % 2 || W % 2)) return sta_fail(STA_E_ARG, "conv3x3_nhwc: up2 needs even H, W");
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const int geo = conv_geo(H, W), ntw = conv_ntw(Cout);
  const int tr = cv_tr(geo), tc = cv_tc(geo);
  CV p{(const char*)x, (const char*)packed_w, (const char*)zeros, out, bias, res, stats, 0, B, H, W, Cin, Cout, up2 ? 1 : 0,
       Cout / cv_part(ntw), W / tc, (H / tr) * (W / tc), 0, 0};
  p.stats_slots = geo == 2 ? 2 : 4 * p.tiles_per_img;
  const long tiles = geo == 2 ? (B + 1) / 2 : (long)B * p.tiles_per_img;
  const long items = tiles * p.parts;
  if (items >= (1l << 30)) return sta_fail(STA_E_UNSUP, "conv3x3_nhwc: too many tiles");
  p.items = (int)items;
  p.xcd_map = tiles % 8 == 0;
  const unsigned grid = (unsigned)(p.items < 256 ? p.items : 256);
  hipStream_t st = (hipStream_t)stream;
  static StaLdsAttr attr[12];
#define STA_CONV_LAUNCH(T, GEO, NTW, A)                                                                                            \
  do {                                                                                                                             \
    if (!attr[A].ensure((const void*)conv3x3_nhwc_kernel<T, GEO, NTW>, cv_lds(NTW, GEO)))                                          \
      return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(conv3x3) failed");                                                        \
    hipLaunchKernelGGL((conv3x3_nhwc_kernel<T, GEO, NTW>), dim3(grid), dim3(64 * CV_NW), cv_lds(NTW, GEO), st, p);                 \
  } while (0)
#define STA_CONV_GEOM(T, A)                                                                                                        \
  do {                                                                                                                             \
    if (geo == 0) { if (ntw == 5) STA_CONV_LAUNCH(T, 0, 5, A); else STA_CONV_LAUNCH(T, 0, 4, A + 1); }                             \
    else if (geo == 1) { if (ntw == 5) STA_CONV_LAUNCH(T, 1, 5, A + 2); else STA_CONV_LAUNCH(T, 1, 4, A + 3); }                    \
    else { if (ntw == 5) STA_CONV_LAUNCH(T, 2, 5, A + 4); else STA_CONV_LAUNCH(T, 2, 4, A + 5); }                                  \
  } while (0)
  if (dtype == STA_BF16) STA_CONV_GEOM(__bf16, 0); else STA_CONV_GEOM(_Float16, 6);
#undef STA_CONV_GEOM
#undef STA_CONV_LAUNCH
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "conv3x3_nhwc launch: %s", hipGetErrorString(e));
}

}  // extern "C"
