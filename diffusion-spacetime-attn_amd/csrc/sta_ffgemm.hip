// sta_ffgemm.hip — the first half of the block's feed-forward at SD-v1 level 0 (C = 320, inner = 1280) as ONE pass:
//
//     h = (y W_v^T + b_v) * gelu(y W_g^T + b_g)          GEGLU.forward (attention.py:42-45) inside FeedForward (:48-69), called at :299
//
// with y = norm3(x) read in QUERY-FRAGMENT order (what sta_to_out_ln_ofrag / sta_add_layernorm_qfrag write). Row-major, this is
// a library GEMM that writes the [R][2560] projection (1.34 GB at 32 images) followed by the GEGLU pass that reads it back and
// writes [R][1280]: here the projection never exists in HBM — the accumulators of a value tile pair and of its gate tile pair
// meet in registers and leave as 8 finished channels per lane.
//
// Rows are the MFMA columns: Out^T = W' y^T. B operand = one 1-KiB fragment of y per k-step (lane (g, c): channels 32 s + 8 g .. + 7
// of row c), 2 x 10 of them per wave (32 rows) held for a whole pass; A operand = proj.weight [2560][320] re-laid out once per
// model: 40 sub-chunks of 40 fragments = {value tiles 2v, 2v+1; gate tiles 2v, 2v+1} x 10 k-steps, rows permuted
// (sigma(2v + t, rho) = 32 v + 8 (rho >> 2) + 4 t + (rho & 3)) so that a lane's two value tiles are 8 CONSECUTIVE channels.
// The 1.6 MB of weight is streamed through a 2-slot LDS ring by LDS-DMA, one 40-KiB sub-chunk ahead, once per 256-row pass
// (8 waves x 32 rows; L2-resident). Per sub-chunk a wave issues 80 MFMAs behind 40 operand reads (each serves both of its
// 16-row items) and finishes 2 x 8 channels per lane with the exact-erf GELU of csrc/sta_unet.hip.
//
// Roofline: MFMA (2 * 320 * 2560 flop per row = 1.64 MFLOP against 640 + 2560 bytes: 512 flop/B).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sta_xattn.h"
#include "sta_unet.h"
#include "sta_internal.h"
#include "sta_xattn_dev.h"

namespace {

constexpr int FF_C = 320, FF_INNER = 1280;
constexpr int FF_NKS = FF_C / 32;              // 10 k-steps
constexpr int FF_NSC = FF_INNER / 32;          // 40 sub-chunks (one pair of value tiles + its pair of gate tiles each)
constexpr int FF_SC_FR = 4 * FF_NKS;           // 40 fragments per sub-chunk
constexpr int FF_NW = 8;
constexpr int FF_PER = FF_SC_FR / FF_NW;       // 5 LDS-DMA instructions per wave per sub-chunk
constexpr int FF_SLOT = FF_SC_FR * FRAG;       // 40 KiB
constexpr int FF_TAB = 2 * FF_INNER * 2;       // bias (value | gate) as 16-bit behind the ring
constexpr int FF_LDS = 2 * FF_SLOT + FF_TAB;

// Exact-erf GELU, the formula of csrc/sta_unet.hip::gelu_erf (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 on erf) with the
// constants folded and the sign handled without a select — gelu(g) = max(g, 0) - |g| * (0.5 * (1 - erf(|g| / sqrt 2))):
// a reciprocal, an exp2 and 11 plain vector instructions per value instead of 16 (|g| is an operand modifier): this kernel's
// vector-issue port is as loaded as its matrix pipe. Max |difference| to the fp64 GELU on [-12, 12]: 5.2e-7.
__device__ __forceinline__ float ff_gelu_erf(float g) {
  const float a = fabsf(g);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752f, a, 1.0f));
  float p = __builtin_fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = __builtin_fmaf(p, t, 0.5f * 1.421413741f);
  p = __builtin_fmaf(p, t, 0.5f * -0.284496736f);
  p = __builtin_fmaf(p, t, 0.5f * 0.254829592f);
  const float e = (p * t) * __builtin_amdgcn_exp2f((-0.5f * 1.4426950408889634f) * g * g);      // 0.5 * (1 - erf(|g| / sqrt 2))
  return __builtin_fmaf(-a, e, fmaxf(g, 0.f));
}

// proj.weight [2 * inner][C] (rows 0 .. inner-1: value, inner ..: gate) -> [sub-chunk v][part][t][k-step f] fragments:
// lane (g, c) holds W[part * inner + 32 v + 8 (c >> 2) + 4 t + (c & 3)][32 f + 8 g .. + 7]
template <typename T>
__global__ __launch_bounds__(64) void pack_w1_kernel(const T* __restrict__ w, T* __restrict__ packed) {
  const int fr = blockIdx.x;                   // ((v * 2 + part) * 2 + t) * NKS + f
  const int f = fr % FF_NKS, t = (fr / FF_NKS) & 1, part = (fr / (2 * FF_NKS)) & 1, v = fr / (4 * FF_NKS);
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int row = part * FF_INNER + 32 * v + 8 * (c >> 2) + 4 * t + (c & 3);
  const typename Tr<T>::V8 x = *(const typename Tr<T>::V8*)(w + (size_t)row * FF_C + 32 * f + 8 * g);
  *(typename Tr<T>::V8*)(packed + (size_t)fr * (FRAG / 2) + lane * 8) = x;
}

struct FF {
  const char* y;        // norm3 output, query-fragment order [R / 16][10][1 KiB]
  const char* w;        // packed proj.weight
  const void* bias;     // [2 * inner] or null
  void* h;              // [R][inner] row-major, or fragment order (h_frag)
  long R;
  int h_frag;           // 1: h leaves as [R / 16][inner / 32] fragments of 1 KiB (lane (g, c): channels 32 s + 8 g .. + 7 of row c) for sta_ff_out_res_hfrag
};

template <typename T>
__global__ __launch_bounds__(64 * FF_NW, 2) void ff_geglu_qfrag_kernel(const FF p) {
  using V8 = typename Tr<T>::V8;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  char* ring = smem;
  T* tab = (T*)(smem + 2 * FF_SLOT);
  for (int i = threadIdx.x; i < 2 * FF_INNER; i += 64 * FF_NW) tab[i] = p.bias ? ((const T*)p.bias)[i] : (T)0.0f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t w_srd = make_srd(p.w, (unsigned)(FF_NSC * FF_SC_FR * FRAG));
  const unsigned lane16 = (unsigned)lane * 16u;
  auto stage = [&](int sc, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < FF_PER; ++i) {
      const int f = wv + FF_NW * i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (__attribute__((address_space(3))) void*)(ring + slot * FF_SLOT + f * FRAG), 16, lane16,
                                               (unsigned)((sc * FF_SC_FR + f) * FRAG), 0, 0);
    }
  };
  const size_t ybytes = (size_t)p.R * FF_C * sizeof(T);
  const __amdgpu_buffer_rsrc_t y_srd = make_srd(p.y, (unsigned)ybytes);
  const __amdgpu_buffer_rsrc_t h_srd = make_srd(p.h, (unsigned)((size_t)p.R * FF_INNER * sizeof(T)));
  const long nblk = (p.R + 32 * FF_NW - 1) / (32 * FF_NW);
  const char* lbase = ring + lane * 16;
  stage(0, 0);
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long row0 = (blk * FF_NW + wv) * 32;            // this wave's two 16-row items; R % 16 == 0
    V8 b[2][FF_NKS];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const long r0 = row0 + 16 * it;
      const unsigned vo = r0 < p.R ? (unsigned)(r0 * FF_C * (long)sizeof(T)) + lane16 : 0xfffffff0u;
#pragma unroll
      for (int f = 0; f < FF_NKS; ++f) b[it][f] = srd_load16<V8>(y_srd, vo, 1024u * f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // y fragments + the sub-chunk 0 DMA (+ the previous pass's last stores)
    auto sub = [&](auto slot_tag, const int sc) __attribute__((always_inline)) {
      constexpr int SLOT = decltype(slot_tag)::value;
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");    // this sub-chunk's DMA landed; the previous epilogue's two stores may fly
      __builtin_amdgcn_s_barrier();
      stage(sc + 1 < FF_NSC ? sc + 1 : 0, SLOT ^ 1);      // next sub-chunk (sub-chunk 0 of the next pass behind the last one)
      const V8* fr = (const V8*)(lbase + SLOT * FF_SLOT);
      f32x4 acc[2][2][2];                                   // [part][t][item]
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        V8 wa[2][FF_NKS];
#pragma unroll
        for (int f = 0; f < FF_NKS; ++f) {
          wa[0][f] = fr[((part * 2 + 0) * FF_NKS + f) * 64];
          wa[1][f] = fr[((part * 2 + 1) * FF_NKS + f) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a01 = a00, a10 = a00, a11 = a00;
#pragma unroll
        for (int f = 0; f < FF_NKS; ++f) {
          a00 = Tr<T>::mfma(wa[0][f], b[0][f], a00);
          a01 = Tr<T>::mfma(wa[0][f], b[1][f], a01);
          a10 = Tr<T>::mfma(wa[1][f], b[0][f], a10);
          a11 = Tr<T>::mfma(wa[1][f], b[1][f], a11);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[part][0][0] = a00; acc[part][0][1] = a01; acc[part][1][0] = a10; acc[part][1][1] = a11;
      }
      // lane (g, c): tiles t = 0, 1, registers r -> channels 32 sc + 8 g + 4 t + r of row c: 8 consecutive channels per item
      const V8 bv = *(const V8*)(tab + 32 * sc + 8 * g), bg = *(const V8*)(tab + FF_INNER + 32 * sc + 8 * g);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const long row = row0 + 16 * it + c16;
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float val = acc[0][e >> 2][it][e & 3] + (float)bv[e];
          const float gate = acc[1][e >> 2][it][e & 3] + (float)bg[e];
          o[e] = (T)(val * ff_gelu_erf(gate));
        }
        // ALWAYS two stores per sub-chunk (rows past R: an offset the descriptor drops): the counted vmcnt above relies on it.
        // Fragment order: the lane's 8 channels ARE lane 16 g + c of fragment `sc` of the item's group — one contiguous KiB per store
        const unsigned ho = row0 + 16 * it >= p.R ? 0xfffffff0u
                          : p.h_frag ? (unsigned)((row0 + 16 * it) * FF_INNER * (long)sizeof(T)) + (unsigned)sc * (unsigned)FRAG + lane16
                                     : (unsigned)(row * FF_INNER * (long)sizeof(T)) + (unsigned)(32 * sc + 8 * g) * (unsigned)sizeof(T);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), h_srd, ho, 0, 0);
      }
    };
    for (int sp = 0; sp < FF_NSC; sp += 2) {
      sub(std::integral_constant<int, 0>{}, sp);
      sub(std::integral_constant<int, 1>{}, sp + 1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// -------------------------------------------------------------------------------------------------------------------------------
// The second half: out = x + h W2^T + b2   (FeedForward's output Linear, attention.py:66-69, + the block's last residual, :299),
// h read in the fragment order the kernel above writes. Same structure as csrc/sta_rowgemm.hip (a wave owns 16 rows and all 20
// output row tiles; the weight streamed through a 2-slot LDS ring in chunks of two row tiles x 10 k-steps), with the reduction
// dimension (1280) walked in four k-chunks of 320: the ten B fragments of the NEXT k-chunk are requested during the last two weight
// chunks of the current one. Roofline: HBM (2560 + 640 + 640 bytes per row against 2 * 1280 * 320 flop: 213 flop/B).
// -------------------------------------------------------------------------------------------------------------------------------
constexpr int F2_NKC = FF_INNER / FF_C;        // 4 k-chunks of 320
constexpr int F2_NRT = FF_C / 16;              // 20 output row tiles
constexpr int F2_NCH = F2_NRT / 2;             // 10 weight chunks (two row tiles) per k-chunk
constexpr int F2_CH_FR = 2 * FF_NKS;           // 20 fragments per chunk
constexpr int F2_PER = (F2_CH_FR + FF_NW - 1) / FF_NW;     // 3 (4 padding copies)
constexpr int F2_SLOT = F2_PER * FF_NW * FRAG; // 24 KiB
constexpr int F2_LDS = 2 * F2_SLOT + FF_C * 2;

// net[2].weight [C][inner] -> [k-chunk kc][chunk v][t][k-step f] fragments: lane (g, c) holds
// W[32 v + 8 (c >> 2) + 4 t + (c & 3)][320 kc + 32 f + 8 g .. + 7]
template <typename T>
__global__ __launch_bounds__(64) void pack_w2_kernel(const T* __restrict__ w, T* __restrict__ packed) {
  const int fr = blockIdx.x;                   // ((kc * NCH + v) * 2 + t) * NKS + f
  const int f = fr % FF_NKS, t = (fr / FF_NKS) & 1, v = (fr / (2 * FF_NKS)) % F2_NCH, kc = fr / (2 * FF_NKS * F2_NCH);
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int row = 32 * v + 8 * (c >> 2) + 4 * t + (c & 3);
  const typename Tr<T>::V8 x = *(const typename Tr<T>::V8*)(w + (size_t)row * FF_INNER + FF_C * kc + 32 * f + 8 * g);
  *(typename Tr<T>::V8*)(packed + (size_t)fr * (FRAG / 2) + lane * 8) = x;
}

struct F2 {
  const char* h;        // fragment order [R / 16][40][1 KiB]
  const char* w;        // packed net[2].weight
  const void* bias;     // [C] or null
  const void* x;        // [R][C] residual
  void* out;            // [R][C]
  long R;
};

template <typename T>
__global__ __launch_bounds__(64 * FF_NW, 2) void ff_out_res_hfrag_kernel(const F2 p) {
  using V8 = typename Tr<T>::V8;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  char* ring = smem;
  T* tab = (T*)(smem + 2 * F2_SLOT);
  for (int i = threadIdx.x; i < FF_C; i += 64 * FF_NW) tab[i] = p.bias ? ((const T*)p.bias)[i] : (T)0.0f;
  const __amdgpu_buffer_rsrc_t w_srd = make_srd(p.w, (unsigned)(F2_NKC * F2_NCH * F2_CH_FR * FRAG));
  const unsigned lane16 = (unsigned)lane * 16u;
  auto stage = [&](int chunk, int slot) __attribute__((always_inline)) {      // chunk = kc * NCH + ch, 0 .. 39
#pragma unroll
    for (int i = 0; i < F2_PER; ++i) {
      const int f = wv + FF_NW * i;
      const int fs = f < F2_CH_FR ? f : 0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (__attribute__((address_space(3))) void*)(ring + slot * F2_SLOT + f * FRAG), 16, lane16,
                                               (unsigned)((chunk * F2_CH_FR + fs) * FRAG), 0, 0);
    }
  };
  const __amdgpu_buffer_rsrc_t h_srd = make_srd(p.h, (unsigned)((size_t)p.R * FF_INNER * sizeof(T)));
  const long nblk = (p.R + 16 * FF_NW - 1) / (16 * FF_NW);
  auto h_off = [&](long blk) -> unsigned {
    const long row0 = (blk * FF_NW + wv) * 16;
    return (blk < nblk && row0 < p.R) ? (unsigned)(row0 * FF_INNER * (long)sizeof(T)) + lane16 : 0xfffffff0u;
  };
  long blk = blockIdx.x;
  V8 b[FF_NKS], bn[FF_NKS];
  stage(0, 0);
  {
    const unsigned vo = h_off(blk);
#pragma unroll
    for (int f = 0; f < FF_NKS; ++f) b[f] = srd_load16<V8>(h_srd, vo, 1024u * f);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const char* lbase = ring + lane * 16;
  for (; blk < nblk; blk += gridDim.x) {
    f32x4 acc[F2_NRT];
#pragma unroll
    for (int u = 0; u < F2_NRT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < F2_NKC; ++kc) {
      auto chunk = [&](auto ch_tag) __attribute__((always_inline)) {
        constexpr int CH = decltype(ch_tag)::value;
        constexpr int SLOT = CH & 1;
        // VMEM operations younger than this chunk's DMA that may stay in flight: the previous pass's 10 stores (chunk 0 of k-chunk 0),
        // the next k-chunk's ten B fragments (chunk 9: requested during chunk 8, behind that chunk's DMA)
        if constexpr (CH == 0) { if (kc == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else if constexpr (CH == F2_NCH - 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage((kc * F2_NCH + CH + 1) % (F2_NKC * F2_NCH), SLOT ^ 1);
        if constexpr (CH == F2_NCH - 2) {      // B fragments of the next k-chunk (of the next pass behind the last one)
          const bool last = kc == F2_NKC - 1;
          const unsigned vo = h_off(last ? blk + gridDim.x : blk);
          const unsigned so = last ? 0u : (unsigned)(kc + 1) * (unsigned)(FF_NKS * FRAG);
#pragma unroll
          for (int f = 0; f < FF_NKS; ++f) bn[f] = srd_load16<V8>(h_srd, vo, so + 1024u * f);
        }
        const V8* fr = (const V8*)(lbase + SLOT * F2_SLOT);
        f32x4 a0 = acc[2 * CH], a1 = acc[2 * CH + 1];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          V8 wa[2][FF_NKS / 2];
#pragma unroll
          for (int f = 0; f < FF_NKS / 2; ++f) {
            wa[0][f] = fr[(half * (FF_NKS / 2) + f) * 64];
            wa[1][f] = fr[(FF_NKS + half * (FF_NKS / 2) + f) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int f = 0; f < FF_NKS / 2; ++f) {
            a0 = Tr<T>::mfma(wa[0][f], b[half * (FF_NKS / 2) + f], a0);
            a1 = Tr<T>::mfma(wa[1][f], b[half * (FF_NKS / 2) + f], a1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        acc[2 * CH] = a0;
        acc[2 * CH + 1] = a1;
      };
      chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{});
      chunk(std::integral_constant<int, 3>{}); chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{});
      chunk(std::integral_constant<int, 6>{}); chunk(std::integral_constant<int, 7>{}); chunk(std::integral_constant<int, 8>{});
      chunk(std::integral_constant<int, 9>{});
#pragma unroll
      for (int f = 0; f < FF_NKS; ++f) b[f] = bn[f];
    }
    // ---- epilogue: + bias + residual; ALWAYS ten stores (rows past R dropped by the descriptor): the counted vmcnt relies on it
    const long row0 = (blk * FF_NW + wv) * 16;
    const long row = row0 + c16;
    const bool ok = row0 < p.R;
    const T* xr = (const T*)p.x + (ok ? row : 0) * FF_C + 8 * g;
    const __amdgpu_buffer_rsrc_t o_srd = make_srd(p.out, (unsigned)((size_t)p.R * FF_C * sizeof(T)));
#pragma unroll
    for (int v = 0; v < F2_NCH; ++v) {
      const V8 xv = *(const V8*)(xr + 32 * v);
      const V8 bs = *(const V8*)(tab + 32 * v + 8 * g);
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (T)(acc[2 * v + (e >> 2)][e & 3] + (float)bs[e] + (float)xv[e]);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), o_srd,
                                             ok ? (unsigned)(row * FF_C * (long)sizeof(T)) + (unsigned)(32 * v + 8 * g) * (unsigned)sizeof(T) : 0xfffffff0u, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

extern "C" {

size_t sta_ff_out_packed_w_bytes(int C, int inner) {
  return (C == FF_C && inner == FF_INNER) ? (size_t)F2_NKC * F2_NCH * F2_CH_FR * FRAG : 0;
}

int sta_ff_out_pack_w(const void* w, void* packed, int C, int inner, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!w || !packed) return sta_fail(STA_E_ARG, "null pointer");
  if (sta_ff_out_packed_w_bytes(C, inner) == 0) return sta_fail(STA_E_UNSUP, "fused feed-forward output: C = 320, inner = 1280 only (C=%d inner=%d)", C, inner);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const unsigned nfr = F2_NKC * F2_NCH * F2_CH_FR;
  if (dtype == STA_BF16) hipLaunchKernelGGL(pack_w2_kernel<__bf16>, dim3(nfr), dim3(64), 0, st, (const __bf16*)w, (__bf16*)packed);
  else hipLaunchKernelGGL(pack_w2_kernel<_Float16>, dim3(nfr), dim3(64), 0, st, (const _Float16*)w, (_Float16*)packed);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "pack_w2 launch: %s", hipGetErrorString(e));
}

int sta_ff_out_res_hfrag(const void* h_frag, const void* packed_w, const void* bias, const void* x, void* out, long R, int C, int inner,
                         int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!h_frag || !packed_w || !x || !out) return sta_fail(STA_E_ARG, "null pointer");
  if (sta_ff_out_packed_w_bytes(C, inner) == 0) return sta_fail(STA_E_UNSUP, "fused feed-forward output: C = 320, inner = 1280 only (C=%d inner=%d)", C, inner);
  if (R <= 0 || R % 16) return sta_fail(STA_E_ARG, "ff_out_res_hfrag: R=%ld (need a positive multiple of 16 rows)", R);
  if ((size_t)R * inner * 2 >= 0xfffffff0ull) return sta_fail(STA_E_UNSUP, "activations must stay below 4 GiB (R=%ld)", R);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  F2 p{(const char*)h_frag, (const char*)packed_w, bias, x, out, R};
  const long nblk = (R + 16 * FF_NW - 1) / (16 * FF_NW);
  const unsigned grid = (unsigned)(nblk < 256 ? nblk : 256);
  hipStream_t st = (hipStream_t)stream;
  static StaLdsAttr attr_b, attr_h;
  if (dtype == STA_BF16) {
    if (!attr_b.ensure((const void*)ff_out_res_hfrag_kernel<__bf16>, F2_LDS)) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(ff_out) failed");
    hipLaunchKernelGGL(ff_out_res_hfrag_kernel<__bf16>, dim3(grid), dim3(64 * FF_NW), F2_LDS, st, p);
  } else {
    if (!attr_h.ensure((const void*)ff_out_res_hfrag_kernel<_Float16>, F2_LDS)) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(ff_out) failed");
    hipLaunchKernelGGL(ff_out_res_hfrag_kernel<_Float16>, dim3(grid), dim3(64 * FF_NW), F2_LDS, st, p);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "ff_out_res_hfrag launch: %s", hipGetErrorString(e));
}

size_t sta_ff_geglu_packed_w_bytes(int C, int inner) {
  return (C == FF_C && inner == FF_INNER) ? (size_t)FF_NSC * FF_SC_FR * FRAG : 0;
}

int sta_ff_geglu_pack_w(const void* w, void* packed, int C, int inner, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!w || !packed) return sta_fail(STA_E_ARG, "null pointer");
  if (sta_ff_geglu_packed_w_bytes(C, inner) == 0) return sta_fail(STA_E_UNSUP, "fused GEGLU projection: C = 320, inner = 1280 only (C=%d inner=%d)", C, inner);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16) hipLaunchKernelGGL(pack_w1_kernel<__bf16>, dim3(FF_NSC * FF_SC_FR), dim3(64), 0, st, (const __bf16*)w, (__bf16*)packed);
  else hipLaunchKernelGGL(pack_w1_kernel<_Float16>, dim3(FF_NSC * FF_SC_FR), dim3(64), 0, st, (const _Float16*)w, (_Float16*)packed);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "pack_w1 launch: %s", hipGetErrorString(e));
}

int sta_ff_geglu_qfrag(const void* y_qfrag, const void* packed_w, const void* bias, void* h, long R, int C, int inner, int h_frag,
                       int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!y_qfrag || !packed_w || !h) return sta_fail(STA_E_ARG, "null pointer");
  if (sta_ff_geglu_packed_w_bytes(C, inner) == 0) return sta_fail(STA_E_UNSUP, "fused GEGLU projection: C = 320, inner = 1280 only (C=%d inner=%d)", C, inner);
  if (R <= 0 || R % 16) return sta_fail(STA_E_ARG, "ff_geglu_qfrag: R=%ld (need a positive multiple of 16 rows)", R);
  if ((size_t)R * inner * 2 >= 0xfffffff0ull) return sta_fail(STA_E_UNSUP, "activations must stay below 4 GiB (R=%ld)", R);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  FF p{(const char*)y_qfrag, (const char*)packed_w, bias, h, R, h_frag ? 1 : 0};
  const long nblk = (R + 32 * FF_NW - 1) / (32 * FF_NW);
  const unsigned grid = (unsigned)(nblk < 256 ? nblk : 256);
  hipStream_t st = (hipStream_t)stream;
  static StaLdsAttr attr_b, attr_h;
  if (dtype == STA_BF16) {
    if (!attr_b.ensure((const void*)ff_geglu_qfrag_kernel<__bf16>, FF_LDS)) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(ff_geglu) failed");
    hipLaunchKernelGGL(ff_geglu_qfrag_kernel<__bf16>, dim3(grid), dim3(64 * FF_NW), FF_LDS, st, p);
  } else {
    if (!attr_h.ensure((const void*)ff_geglu_qfrag_kernel<_Float16>, FF_LDS)) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(ff_geglu) failed");
    hipLaunchKernelGGL(ff_geglu_qfrag_kernel<_Float16>, dim3(grid), dim3(64 * FF_NW), FF_LDS, st, p);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "ff_geglu_qfrag launch: %s", hipGetErrorString(e));
}

}  // extern "C"
