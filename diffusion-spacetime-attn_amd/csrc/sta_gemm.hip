// sta_gemm.hip — the Linear layers / 1x1 convolutions of the UNet's transformer blocks and ResBlock skips as ONE row GEMM kernel
//
//     out[r][n] = sum_k x[r][k] * w[n][k] + bias[n] + res[r][n]            x: [R][K] rows (tokens of [B, N, C], or NHWC pixels)
//
// (reference: CrossAttention.to_q / to_k / to_v / to_out, attention.py:158-173, :178-183, :215; FeedForward / GEGLU projections :42-69;
// SpatialTransformer.proj_in / proj_out :322-333; ResBlock.skip_connection, openaimodel.py:196-206). Through hipBLASLt these shapes —
// 65 536 .. 262 144 rows against K = 320 .. 1280 — run at 0.35 - 0.75 PFLOP/s and 1.7 - 2.2 TB/s (`tools/gemm_census.py`): short
// reduction loops and outputs as large as the inputs. This is csrc/sta_conv.hip without the taps: rows are the MFMA columns,
// Out^T[n][r] = W . X^T (v_mfma_f32_16x16x32); a workgroup (8 waves) owns 256 rows x 160 (or 128) output columns, a wave 64 rows x 80
// (64) columns; the row tile crosses LDS once per 64-channel step (LDS-DMA with per-lane source addresses, 16-byte chunks swizzled so
// that every B-operand read is bank-conflict-free), the weight — re-laid out once per model into 1-KiB A fragments
// [part][step][k-chunk][tile] — is streamed through a 2-slot ring, one barrier per step (40 MFMAs per wave), the next step's DMA
// issued behind the first k-chunk's MFMAs; persistent workgroups, the parts of a row tile back to back on one XCD; epilogue
// + bias + res with 8-byte stores. Roofline: HBM at K = 320 (160 flop/B), MFMA above.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "sta_xattn.h"
#include "sta_unet.h"
#include "sta_internal.h"
#include "sta_xattn_dev.h"

namespace {

constexpr int GM_NW = 8;
constexpr int GM_ROWS = 256;                                                // rows per workgroup
constexpr int gm_part(int ntw) { return 32 * ntw; }
constexpr int gm_nt(int ntw) { return 2 * ntw; }
constexpr int gm_wfr(int ntw) { return 2 * gm_nt(ntw); }                    // 20 / 16 weight fragments per step (two k-chunks)
constexpr int gm_wper(int ntw) { return (gm_wfr(ntw) + GM_NW - 1) / GM_NW; }     // 3 / 2
constexpr int gm_wslot(int ntw) { return gm_wper(ntw) * GM_NW * FRAG; }     // 24 / 16 KiB
constexpr int GM_XPER = 2 * (GM_ROWS / 16) / GM_NW;                         // 4 input pieces (16 rows x 32 channels) per wave per step
constexpr int GM_XBUF = 2 * GM_ROWS * 64;                                   // 32 KiB: [k-chunk][row][64 B]
constexpr int gm_lds(int ntw) { return 2 * gm_wslot(ntw) + 2 * GM_XBUF; }   // 112 / 96 KiB

// w[n][k] (element (n, k) at n * sn + k * sk) -> fragments [part][step][kci][tile t]: lane (g, c) holds W[16 nt part + 16 t + c][64 step + 32 kci + 8 g .. + 7]
template <typename T>
__global__ __launch_bounds__(64) void pack_gemm_w_kernel(const T* __restrict__ w, long sn, long sk, T* __restrict__ packed, int nsteps, int nt) {
  const int fr = blockIdx.x;                   // ((part * nsteps + step) * 2 + kci) * nt + t
  const int t = fr % nt, kci = (fr / nt) & 1, step = (fr / (2 * nt)) % nsteps, part = fr / (2 * nt * nsteps);
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const T* src = w + (size_t)(16 * nt * part + 16 * t + c) * sn + (size_t)(64 * step + 32 * kci + 8 * g) * sk;
  typename Tr<T>::V8 x;
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = src[j * sk];
  *(typename Tr<T>::V8*)(packed + (size_t)fr * (FRAG / 2) + lane * 8) = x;
}

__device__ __forceinline__ float gm_row16_sum(float x) {     // sum over the 16 lanes of a DPP row, result in every lane
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));
  return x;
}

struct GM {
  const char* x;        // [R][K], or [R][Ka] when xb is set
  const char* xb;       // null, or [R][K - Ka]: the input is the column concatenation [x | xb], read in place (Ka % 64 == 0)
  int Ka;
  const char* w;        // packed weights
  const char* zeros;    // >= 2 * K bytes of zeros (rows past R)
  void* out;            // [R][N]
  const void* bias;     // [N] or null
  const void* res;      // [R][N] or null
  float* stats;         // null, or [R / rows_img + 1][rows_img / 256 * 4][N][2] fp32: per (image, slot) partial sums / sums of squares of the stored values
  long rows_img;
  long R;
  int K, N, parts, items, xcd_map;
  long tiles;
};

template <typename T, int NTW>
__global__ __launch_bounds__(64 * GM_NW, 1) void gemm_rows_kernel(const GM p) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  constexpr int PART = gm_part(NTW), NT = gm_nt(NTW), WFR = gm_wfr(NTW), WPER = gm_wper(NTW), WSLOT = gm_wslot(NTW);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int pq = wv & 3, ch = wv >> 2;                     // row quarter (4 groups of 16 rows), column half
  char* xring = smem;
  char* wring = smem + 2 * GM_XBUF;
  const int nsteps = p.K >> 6;
  const __amdgpu_buffer_rsrc_t w_srd = make_srd(p.w, (unsigned)((size_t)p.parts * nsteps * WFR * FRAG));
  const unsigned lane16 = (unsigned)lane * 16u;

  auto item_tile = [&](int it, long& tile, int& part) {
    if (p.xcd_map) {
      const int j = it >> 3;
      tile = (long)(j / p.parts) * 8 + (it & 7);
      part = j % p.parts;
    } else {
      tile = it / p.parts;
      part = it - (int)tile * p.parts;
    }
  };
  // input piece pc of a step: k-chunk pc >> 4, rows 16 (pc & 15) .. + 15 of the tile; this lane: row + (lane >> 2), LDS slot lane & 3
  // (second = true: the first step that reads the second tensor of a concatenated input)
  const int nsteps_a = p.xb ? p.Ka >> 6 : nsteps;
  auto in_ptr = [&](long tile, int pc, bool second) -> const char* {
    const int rl = 16 * (pc & 15) + (lane >> 2);
    const long row = tile * GM_ROWS + rl;
    const int chunk = (lane & 3) ^ (((rl >> 2) & 1) << 1);
    const char* base = second ? p.xb + (size_t)row * (p.K - p.Ka) * sizeof(T) : p.x + (size_t)row * (p.xb ? p.Ka : p.K) * sizeof(T);
    return row < p.R ? base + (pc >> 4) * 64 + chunk * 16 : p.zeros + (pc >> 4) * 64 + chunk * 16;
  };
  auto stage_w = [&](int part, int step, int slot) __attribute__((always_inline)) {
    const unsigned base = (unsigned)((part * nsteps + step) * WFR) * (unsigned)FRAG;
#pragma unroll
    for (int i = 0; i < WPER; ++i) {
      const int f = wv + GM_NW * i;
      const int fs = f < WFR ? f : 0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (__attribute__((address_space(3))) void*)(wring + slot * WSLOT + f * FRAG), 16, lane16,
                                               base + (unsigned)fs * (unsigned)FRAG, 0, 0);
    }
  };
  auto stage_x = [&](const char* src, int pc, int buf) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(xring + buf * GM_XBUF + pc * FRAG), 16, 0, 0);
  };
  // B operand of row group q: row 64 pq + 16 q + c16 of the tile, chunk g at its swizzled slot
  unsigned seg_e[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rl = 64 * pq + 16 * q + c16;
    seg_e[q] = (unsigned)(rl * 64 + ((g ^ (((rl >> 2) & 1) << 1)) << 4));
  }

  int it = blockIdx.x;
  if (it >= p.items) return;
  long tile;
  int part;
  item_tile(it, tile, part);
  const char* xp[GM_XPER];                                 // this wave's pieces (wv, wv + 8, wv + 16, wv + 24) of the NEXT step
#pragma unroll
  for (int i = 0; i < GM_XPER; ++i) xp[i] = in_ptr(tile, wv + GM_NW * i, false);
  stage_w(part, 0, 0);
#pragma unroll
  for (int i = 0; i < GM_XPER; ++i) { stage_x(xp[i], wv + GM_NW * i, 0); xp[i] += 128; }

  const size_t obytes = (size_t)p.R * p.N * sizeof(T);
  const __amdgpu_buffer_rsrc_t o_srd = make_srd(p.out, (unsigned)(obytes < 0xfffffff0ull ? obytes : 0xfffffff0ull));
  const __amdgpu_buffer_rsrc_t r_srd = make_srd(p.res ? p.res : p.out, (unsigned)(obytes < 0xfffffff0ull ? obytes : 0xfffffff0ull));
  bool first_of_tile = false;
  int par = 0;                                             // ring slot of the current tile's first step

  while (true) {
    f32x4 acc[4][NTW];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nit = it + gridDim.x;
    long ntile = 0;
    int npart = 0;
    const bool more = nit < p.items;
    if (more) item_tile(nit, ntile, npart);

    auto step = [&](auto slot_tag, const int st) __attribute__((always_inline)) {
      constexpr int SL = decltype(slot_tag)::value;
      if (first_of_tile) {
        if (p.stats) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NTW + 2 * NTW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NTW) : "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      first_of_tile = false;
      __builtin_amdgcn_s_barrier();
      const bool last = st + 1 == nsteps;
      auto stage_next = [&]() __attribute__((always_inline)) {
        if (!last) stage_w(part, st + 1, SL ^ 1);
        else if (more) stage_w(npart, 0, SL ^ 1);
        if (!last || more) {
#pragma unroll
          for (int i = 0; i < GM_XPER; ++i) {
            if (last) xp[i] = in_ptr(ntile, wv + GM_NW * i, false);
            else if (st + 1 == nsteps_a) xp[i] = in_ptr(tile, wv + GM_NW * i, true);      // the concatenated input's second tensor starts
            stage_x(xp[i], wv + GM_NW * i, SL ^ 1);
            xp[i] += 128;
          }
        }
      };
      const char* xb = xring + SL * GM_XBUF;
      const V8* wf = (const V8*)(wring + SL * WSLOT + lane * 16) + (NTW * ch) * 64;
      V8 a[NTW], b[4];
      auto load_kc = [&](int kci, V8 (&aa)[NTW], V8 (&bb)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bb[q] = *(const V8*)(xb + kci * (GM_ROWS * 64) + seg_e[q]);
#pragma unroll
        for (int t = 0; t < NTW; ++t) aa[t] = wf[(kci * NT + t) * 64];
      };
      load_kc(0, a, b);
#pragma unroll
      for (int kci = 0; kci < 2; ++kci) {
        V8 an[NTW], bn[4];
        if (kci < 1) load_kc(kci + 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q][t] = Tr<T>::mfma(a[t], b[q], acc[q][t]);
        __builtin_amdgcn_sched_barrier(0);
        if (kci == 0) {
          stage_next();
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < NTW; ++t) a[t] = an[t];
#pragma unroll
          for (int q = 0; q < 4; ++q) b[q] = bn[q];
        }
      }
    };
    // the ring slot alternates with every step ACROSS tiles: a tile with an odd number of steps (K = 320, 960) hands the next one
    // the other starting slot — the loop exists once per starting parity
    auto run_tile = [&](auto p0_tag) __attribute__((always_inline)) {
      constexpr int P0 = decltype(p0_tag)::value;
      for (int st = 0; st < nsteps; st += 2) {
        step(std::integral_constant<int, P0>{}, st);
        if (st + 1 < nsteps) step(std::integral_constant<int, P0 ^ 1>{}, st + 1);
      }
    };
    if (par == 0) run_tile(std::integral_constant<int, 0>{});
    else run_tile(std::integral_constant<int, 1>{});
    par ^= nsteps & 1;
    // epilogue: lane (g, c) of tile t holds columns 16 t + 4 g .. + 3 of row c: ALWAYS 4 NTW stores of 8 bytes
    {
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      float bs[NTW][4];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        V4 bv = {};
        if (p.bias) bv = *(const V4*)((const T*)p.bias + part * PART + ch * (16 * NTW) + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[t][r] = (float)bv[r];
      }
      float ssum[NTW][4], ssq[NTW][4];
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[t][r] = 0.f; ssq[t][r] = 0.f; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long row = tile * GM_ROWS + 64 * pq + 16 * q + c16;
        const bool ok = row < p.R;
        const unsigned base = (unsigned)(((size_t)row * p.N + part * PART + ch * (16 * NTW) + 4 * g) * sizeof(T));
        V4 rv[NTW];
        if (p.res) {
#pragma unroll
          for (int t = 0; t < NTW; ++t)
            rv[t] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b64(r_srd, ok ? base + (unsigned)(16 * t * sizeof(T)) : 0xfffffff0u, 0, 0));
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          V4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (T)(acc[q][t][r] + bs[t][r] + (p.res ? (float)rv[t][r] : 0.f));
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), o_srd, ok ? base + (unsigned)(16 * t * sizeof(T)) : 0xfffffff0u, 0, 0);
          if (p.stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = ok ? (float)o[r] : 0.f;
              ssum[t][r] += v;
              ssq[t][r] += v * v;
            }
          }
        }
      }
      if (p.stats) {                                       // partial sums of this wave's 4 x NTW columns: ALWAYS 2 NTW stores per wave
        const long row0 = tile * GM_ROWS;
        const bool live = row0 < p.R;
        const long img = live ? row0 / p.rows_img : p.R / p.rows_img;          // (a tile past R writes the spare image slot)
        const int slot = live ? (int)((row0 - img * p.rows_img) / GM_ROWS) * 4 + pq : pq;
        const int slots = (int)(p.rows_img / GM_ROWS) * 4;
        float* dst = p.stats + (((size_t)img * slots + slot) * p.N + part * PART + ch * (16 * NTW) + 4 * g) * 2;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          f32x4 lo, hi;
          lo[0] = gm_row16_sum(ssum[t][0]); lo[1] = gm_row16_sum(ssq[t][0]); lo[2] = gm_row16_sum(ssum[t][1]); lo[3] = gm_row16_sum(ssq[t][1]);
          hi[0] = gm_row16_sum(ssum[t][2]); hi[1] = gm_row16_sum(ssq[t][2]); hi[2] = gm_row16_sum(ssum[t][3]); hi[3] = gm_row16_sum(ssq[t][3]);
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

}  // namespace

extern "C" {

static int gemm_ntw(int N) { return N > 0 && N % 160 == 0 ? 5 : (N > 0 && N % 128 == 0 ? 4 : 0); }

int sta_linear_rows_supported(long R, int K, int N) {
  if (R <= 0 || K <= 0 || K % 64 || gemm_ntw(N) == 0) return 0;
  if ((size_t)R * N * 2 >= 0xfffffff0ull) return 0;
  return 1;
}

size_t sta_linear_rows_packed_w_bytes(int K, int N) { return (K > 0 && K % 64 == 0 && gemm_ntw(N)) ? (size_t)N * K * 2 : 0; }

int sta_linear_rows_pack_w(const void* w, long sn, long sk, void* packed, int K, int N, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!w || !packed) return sta_fail(STA_E_ARG, "null pointer");
  if (sta_linear_rows_packed_w_bytes(K, N) == 0) return sta_fail(STA_E_UNSUP, "linear_rows: K %% 64 == 0 and N %% 160 == 0 or N %% 128 == 0 (K=%d N=%d)", K, N);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int nsteps = K / 64, nt = gm_nt(gemm_ntw(N));
  const unsigned nfr = (unsigned)(N / (16 * nt)) * nsteps * 2 * nt;
  if (dtype == STA_BF16) hipLaunchKernelGGL(pack_gemm_w_kernel<__bf16>, dim3(nfr), dim3(64), 0, st, (const __bf16*)w, sn, sk, (__bf16*)packed, nsteps, nt);
  else hipLaunchKernelGGL(pack_gemm_w_kernel<_Float16>, dim3(nfr), dim3(64), 0, st, (const _Float16*)w, sn, sk, (_Float16*)packed, nsteps, nt);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "pack_gemm_w launch: %s", hipGetErrorString(e));
}

static int linear_rows_impl(const void* x, const void* xb, int Ka, const void* packed_w, const void* zeros, const void* bias, const void* res,
                            void* out, long R, int K, int N, int dtype, void* stream, float* stats = nullptr, long rows_img = 0) {
  g_sta_err[0] = 0;
  if (!x || !packed_w || !zeros || !out) return sta_fail(STA_E_ARG, "null pointer");
  if (xb && (Ka <= 0 || Ka >= K || Ka % 64)) return sta_fail(STA_E_ARG, "linear_rows_cat: Ka=%d of K=%d (need 0 < Ka < K, Ka %% 64 == 0)", Ka, K);
  if (!sta_linear_rows_supported(R, K, N)) return sta_fail(STA_E_UNSUP, "linear_rows: unsupported shape R=%ld K=%d N=%d", R, K, N);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  const int ntw = gemm_ntw(N);
  if (stats && (rows_img <= 0 || rows_img % GM_ROWS || R % rows_img)) return sta_fail(STA_E_ARG, "linear_rows_stats: rows_per_image=%ld (need a multiple of 256 dividing R=%ld)", rows_img, R);
  GM p{(const char*)x, (const char*)xb, Ka, (const char*)packed_w, (const char*)zeros, out, bias, res, stats, rows_img, R, K, N, N / gm_part(ntw), 0, 0, (R + GM_ROWS - 1) / GM_ROWS};
  const long items = p.tiles * p.parts;
  if (items >= (1l << 30)) return sta_fail(STA_E_UNSUP, "linear_rows: too many tiles");
  p.items = (int)items;
  p.xcd_map = p.tiles % 8 == 0;
  const unsigned grid = (unsigned)(p.items < 256 ? p.items : 256);
  hipStream_t st = (hipStream_t)stream;
  static StaLdsAttr attr[4];
#define STA_GEMM_LAUNCH(T, NTW, A)                                                                                                 \
  do {                                                                                                                             \
    if (!attr[A].ensure((const void*)gemm_rows_kernel<T, NTW>, gm_lds(NTW))) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(linear_rows) failed"); \
    hipLaunchKernelGGL((gemm_rows_kernel<T, NTW>), dim3(grid), dim3(64 * GM_NW), gm_lds(NTW), st, p);                              \
  } while (0)
  if (dtype == STA_BF16) { if (ntw == 5) STA_GEMM_LAUNCH(__bf16, 5, 0); else STA_GEMM_LAUNCH(__bf16, 4, 1); }
  else { if (ntw == 5) STA_GEMM_LAUNCH(_Float16, 5, 2); else STA_GEMM_LAUNCH(_Float16, 4, 3); }
#undef STA_GEMM_LAUNCH
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "linear_rows launch: %s", hipGetErrorString(e));
}

int sta_linear_rows(const void* x, const void* packed_w, const void* zeros, const void* bias, const void* res, void* out, long R, int K, int N,
                    int dtype, void* stream) {
  return linear_rows_impl(x, nullptr, 0, packed_w, zeros, bias, res, out, R, K, N, dtype, stream);
}

int sta_linear_rows_stats(const void* x, const void* packed_w, const void* zeros, const void* bias, const void* res, void* out, float* stats,
                          long rows_per_image, long R, int K, int N, int dtype, void* stream) {
  if (!stats) { g_sta_err[0] = 0; return sta_fail(STA_E_ARG, "null pointer"); }
  return linear_rows_impl(x, nullptr, 0, packed_w, zeros, bias, res, out, R, K, N, dtype, stream, stats, rows_per_image);
}

int sta_linear_rows_cat(const void* xa, const void* xb, int Ka, const void* packed_w, const void* zeros, const void* bias, const void* res,
                        void* out, long R, int K, int N, int dtype, void* stream) {
  if (!xb) { g_sta_err[0] = 0; return sta_fail(STA_E_ARG, "null pointer"); }
  return linear_rows_impl(xa, xb, Ka, packed_w, zeros, bias, res, out, R, K, N, dtype, stream);
}

}  // extern "C"
