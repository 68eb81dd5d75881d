// sta_xattn_proj3.hip — projection-fused forward for HEAD PAIRS, second generation (d = 40: SD-v1 level 0, K <= 2).
//
// Same decomposition as its predecessor (a workgroup = 8 waves x 16 pixels keeps ONE head pair's operands in LDS — the
// pair's Wq fragments and the K / V^T of every context — and walks strided pixel tiles: per tile the projection
// Q^T = Wq_pair y^T of both batch rows, then per head the K+2 attentions and the masked blend), rebuilt around what the
// round-2 instruction stream showed (profiles/r03_level0.md): the waves were parked on LDS latency — every operand
// was requested right before the MFMA that consumed it — and a third of the vector instructions were register moves
// that re-assembled operands from pairs of 8-byte reads.
//
//   * operand images are laid out so that ONE ds_read_b128 (or ds_read_b64) is one MFMA operand, with no padding FLOPs:
//       S^T = K Q^T     per key tile: v_mfma_16x16x32 over 32 head dims + v_mfma_16x16x16 over the remaining 8 (+ 8 zeros)
//       O^T = V^T P^T   per dim tile: 2 x v_mfma_16x16x32 over keys 0..63 + v_mfma_16x16x16 over keys 64..79
//     K row = 96 B: 4 chunks [dims 4g.. | dims 16+4g..] + 4 units (dims 32..39, zeros)  (head B: shifted by the 8 dims
//     that share projection tile 2 with head A — the ZEROS sit in the image, the q operands are plain conversions of
//     the projection accumulators, one small operand serves both heads);  V^T row = 160 B: keys in S^T accumulator
//     order (2 x 64 B) + keys 64..79; row 40 = ones (softmax denominator out of the PV MFMAs).
//     13952 B per (ctx, head): 4 contexts x 2 heads + 50 KiB of Wq = 162816 B of the CU's 163840.
//   * explicit software pipeline: the Wq fragments of k-step s+1 are requested before the MFMAs of step s; the K
//     operands of the NEXT context are requested after this context's S^T MFMAs and land under its softmax, the V^T
//     operands before them; sched_barrier(0) pins the request points (hipcc otherwise sinks every read to its use).
//   * the denominator is broadcast with two permlane swaps instead of an LDS bpermute round trip.
//
// Reference replaced: ldm/modules/attention.py:178 (to_q), :175-197 (the K+2 attentions), :278-294 (masked blend);
// reached through sta_xattn_fwd_proj (sta_xattn_proj.hip dispatches here). tools/emu_pair3.py holds the layout algebra.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "sta_xattn.h"
#include "sta_internal.h"
#include "sta_xattn_dev.h"
#include "sta_xattn_proj3.h"

// (The ablation / timeline build of this kernel that profiles/r03_level0.md and profiles/r04_level0.md quote was a copy of the round-4
// source; it no longer matched this file — optimistic softmax, swizzled operand slots — and was deleted in round 6: git history has it.)

namespace {

using namespace sta_p3;

template <typename T> struct M16;           // the k = 16 MFMA of the same type family
template <> struct M16<_Float16> {
  static __device__ __forceinline__ f32x4 mfma(f16x4 a, f16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};
template <> struct M16<__bf16> {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  static __device__ __forceinline__ f32x4 mfma(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
};

// K, V [n_ctx][M][C] -> [ctx][head][K rows | V^T rows], BLK bytes each (layout: file header, tools/emu_pair3.py)
template <typename T>
__global__ __launch_bounds__(256) void pack_kv_p3_kernel(const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ packed,
                                                         int M, int C, int H) {
  const int ch = blockIdx.x;                  // ctx * H + h
  const int ctx = ch / H, h = ch % H, hp = h & 1;
  T* blk = (T*)((char*)packed + (size_t)ch * BLK);
  const T* kh = k + (size_t)ctx * M * C + h * D;
  const T* vh = v + (size_t)ctx * M * C + h * D;
  for (int i = threadIdx.x; i < BLK / 2; i += blockDim.x) {
    T x = (T)0.0f;
    if (i < KBYTES / 2) {
      const int key = i / (KROW / 2), pos = i % (KROW / 2);
      int dim = -1;
      if (pos < 32) {
        const int g = swz_big(pos >> 3, key), j = pos & 7;            // the 16-byte chunk at slot (pos >> 3) holds lane row g's operand
        dim = hp == 0 ? (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)) : (j < 4 ? 8 + 4 * g + j : 24 + 4 * g + (j - 4));
      } else {
        const int g = swz_small((pos - 32) >> 2, key), j = (pos - 32) & 3;
        if (hp == 0) { if (g < 2) dim = 32 + 4 * g + j; } else { if (g >= 2) dim = 4 * (g - 2) + j; }
      }
      if (key < M && dim >= 0) x = kh[(size_t)key * C + dim];
    } else {
      const int i2 = i - KBYTES / 2;
      const int r = i2 / (VROW / 2), pos = i2 % (VROW / 2);
      const int dim = vrow_dim(r, hp);
      int key;
      if (pos < 64) {
        const int s = pos >> 5, g = swz_big((pos >> 3) & 3, r), j = pos & 7;
        key = 32 * s + 16 * (j >> 2) + 4 * g + (j & 3);
      } else {
        const int g = swz_small((pos - 64) >> 2, r), j = (pos - 64) & 3;
        key = 64 + 4 * g + j;
      }
      if (key < M) x = dim >= 0 ? vh[(size_t)key * C + dim] : (T)1.0f;
    }
    blk[i] = x;
  }
}

struct P3 {
  const void* y;
  const char* wq;        // pair fragments: [pair][nkc][NT], 1 KiB each
  const char* kv;        // [I][K+2][H][BLK]
  const uint8_t* mask;
  const float* coef;
  void* out;
  int N, C, H, M, K, W, tiles, iters;
  float sl2e;
  unsigned* stats;       // null, or the caller's STA_P3_STATS_WORDS words (include/sta_xattn.h: sta_xattn_fwd_proj_ex)
};

template <typename T> struct KFr {            // the K operands of one (context, head): 5 key tiles
  typename Tr<T>::V8 big[NKT];
  typename Tr<T>::V4 sm[NKT];
};

template <typename T>
__device__ __forceinline__ void load_k(KFr<T>& kf, const char* kb, const char* ks) {
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    kf.big[t] = *(const typename Tr<T>::V8*)(kb + t * 16 * KROW);
    kf.sm[t] = *(const typename Tr<T>::V4*)(ks + t * 16 * KROW);
  }
}

// lane (g, c) <- lane (2, c): the denominator sits in lane row 2 (row 40 of O^T = tile 2, row 8). Two row swaps:
// v_permlane32_swap (rows 0,1 <- rows 2,3), then v_permlane16_swap (odd rows <- even rows). Hand-placed wait states: with the
// builtins hipcc let LDS reads stand in for the 2 wait states a swap needs after the VALU write of its operands, and the second
// swap of every context but the first then read stale registers on gfx950 (tools/p3_debug2.py) — the s_nop are inside the string.
//
// What round 4 took for a "code-generation hazard around this statement" (profiles/r04_level0.md section 5: the bf16 out-fragment
// instantiation wrong in registers 0 and 1 of a tile, every source-level cure moving the failure elsewhere) is a hardware rule hipcc
// 7.2 does not know (profiles/r05_hazard_table.md, tools/hazard_probe.py): an MFMA that reads as C the result of an MFMA of ANOTHER
// shape (the k = 16 tail of a chain behind its k = 32 steps) fewer than 5 wait states after it, with no third MFMA between them, takes
// registers 0 and 1 of the tile before they are written. hipcc pads that pair with nothing when the destination is the same, so
// whether a chain was hit depended on what the scheduler happened to put between the two MFMAs (one v_xor in the bf16 out-fragment
// instantiation: wrong; three vector instructions in eight others: right by one state). sta.lib.build() compiles every source to
// assembly, runs sta/isa_lint.py over it and pads what it finds (csrc/.isa_lint.log); all six instantiations are then oracle-exact.
__device__ __forceinline__ float bcast_row2(float x) {
  // Round 6: the two swaps as builtins again. Rounds 3 - 5 kept them inside one asm statement with five `s_nop 1` (10 wait states per
  // context, ~1.5 % of the kernel's issue slots) because "hipcc let LDS reads stand in for the wait states and the second swap read stale
  // registers" — what was really wrong there was the mixed-shape MFMA chain (profiles/r05_hazard_table.md), and the measured table
  // (tools/hazard_probe.py ^perm: vector write -> swap 1 state by ANY filler, swap -> vector read 0, swap -> swap 1) says the compiler's
  // own padding (2 states, filled with whatever it can schedule there) is enough; sta/isa_lint.py re-checks every site.
#ifdef STA_P3_ASM_BCAST     // rounds 3 - 5's form, for same-box A/B builds only (tools/lib_ab.py ... asm=-DSTA_P3_ASM_BCAST)
  unsigned t0, t1;
  asm volatile("s_nop 1\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %2\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\t"
               "v_mov_b32 %0, %1\n\ts_nop 1\n\tv_permlane16_swap_b32 %1, %0\n\ts_nop 1"
               : "=&v"(t0), "=&v"(t1) : "v"(x));
  return __uint_as_float(t1);
#endif
  const unsigned u = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);        // second operand: lane rows (2, 3, 2, 3)
  auto b = __builtin_amdgcn_permlane16_swap(a[1], a[1], false, false);  // first operand: lane rows (2, 2, 2, 2)
  return __uint_as_float(b[0]);
}

// What a head leaves for the stores of one batch row: `main` = the lane's tile-0 | tile-1 outputs (8 consecutive dims, 16 bytes),
// `tail` = its tile-2 outputs (dims 32 + 4g .. +3 in lane rows 0 and 1; rows 2, 3 hold the ones row and padding)
struct OutRow {
  u32x4 main;
  unsigned tail[2];
};
template <typename T>
__device__ __forceinline__ OutRow pack_out(const f32x4 (&a)[3]) {
  typedef __attribute__((ext_vector_type(2))) T T2;
  OutRow r;
  r.main = u32x4{__builtin_bit_cast(unsigned, T2{(T)a[0][0], (T)a[0][1]}), __builtin_bit_cast(unsigned, T2{(T)a[0][2], (T)a[0][3]}),
                 __builtin_bit_cast(unsigned, T2{(T)a[1][0], (T)a[1][1]}), __builtin_bit_cast(unsigned, T2{(T)a[1][2], (T)a[1][3]})};
  r.tail[0] = __builtin_bit_cast(unsigned, T2{(T)a[2][0], (T)a[2][1]});
  r.tail[1] = __builtin_bit_cast(unsigned, T2{(T)a[2][2], (T)a[2][3]});
  return r;
}
// Bytes 64..159 of the pair segment of one output row (bytes 0..63 = head A's `main`, stored when head A is done): one
// v_permlane16_swap per tail dword hands lane row 0 the A tail [own | row 1's] and lane row 1 the B tail; then
//   bytes  64..127: lane row 0 = A tail, rows 1..3 = B main (B dims 0..23)      address = seg + 64 + 16 g
//   bytes 128..159: lane row 0 = B main (B dims 24..31), row 1 = B tail          address = seg + 128 + 16 g, rows 2, 3 off
// `seg` = byte offset of the pair segment of this lane's pixel row in the output buffer; lanes that must not store (pixel
// >= N: `valid` false; lane rows 2, 3 of the last store) get the offset 0xfffffff0, which the descriptor's bounds check drops.
__device__ __forceinline__ void store_pair_rest(__amdgpu_buffer_rsrc_t srd, unsigned seg, bool valid, int g, const unsigned (&tailA)[2], const OutRow& b) {
  auto s0 = __builtin_amdgcn_permlane16_swap(tailA[0], b.tail[0], false, false);
  auto s1 = __builtin_amdgcn_permlane16_swap(tailA[1], b.tail[1], false, false);
  const u32x4 tail = {s0[0], s1[0], s0[1], s1[1]};
  const bool row0 = g == 0;
  const u32x4 v2 = {row0 ? tail[0] : b.main[0], row0 ? tail[1] : b.main[1], row0 ? tail[2] : b.main[2], row0 ? tail[3] : b.main[3]};
  const u32x4 v3 = {row0 ? b.main[0] : tail[0], row0 ? b.main[1] : tail[1], row0 ? b.main[2] : tail[2], row0 ? b.main[3] : tail[3]};
  __builtin_amdgcn_raw_buffer_store_b128(v2, srd, valid ? seg + 64u + 16u * (unsigned)g : 0xfffffff0u, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b128(v3, srd, (valid && g < 2) ? seg + 128u + 16u * (unsigned)g : 0xfffffff0u, 0, 0);
}

template <typename T>
__device__ __forceinline__ typename Tr<T>::V8 cat8(const f32x4& lo, const f32x4& hi) {
  typename Tr<T>::V8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = (T)lo[j]; r[4 + j] = (T)hi[j]; }
  return r;
}
template <typename T>
__device__ __forceinline__ typename Tr<T>::V4 cvt4(const f32x4& a) {
  // two PAIR conversions: written element by element hipcc 7.2 emits, for bf16, four single v_cvt_pk_bf16_f32 + two v_perm_b32
  typedef __attribute__((ext_vector_type(2))) float F2;
  typedef __attribute__((ext_vector_type(2))) T T2;
  typedef __attribute__((ext_vector_type(2))) unsigned U2;
  const T2 lo = __builtin_convertvector(F2{a[0], a[1]}, T2), hi = __builtin_convertvector(F2{a[2], a[3]}, T2);
  return __builtin_bit_cast(typename Tr<T>::V4, U2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)});
}

// The OPTIMISTIC softmax (both 16-bit types since the fp16 window below; first built for bf16: profiles/r05_level0.md: a SIMD's matrix and vector cycles add up in this kernel, and
// 54 of a context's 175 ns of vector work are the running maximum — 9 v_max3, the row butterfly — and the scale-and-subtract FMAs).
// A bf16 P operand has fp32's exponent range, so P = exp2(S) needs no maximum as long as the denominator (the ones row of V^T, summed
// by the PV MFMAs in fp32) stays inside [2^-100, 2^100): |logit| < ~65. The scores arrive in log2 units (scale * log2 e is folded into the Wq
// fragments in LDS, once per workgroup). The denominator's range is checked per context — outside it in any
// lane sends the WAVE through the standard path for that context (K and V^T operands re-read from LDS: exact for any input,
// tests/test_kernel_gpu.py::test_fwd_proj_pair_extreme_logits). fp16: the same with a narrower window (kDenLo / kDenHi).
#ifndef STA_P3_OPTIMISTIC
#define STA_P3_OPTIMISTIC 1      // 0: the standard softmax in both types (same-box A/B builds: tools/asm_patch_ab.py flag:-DSTA_P3_OPTIMISTIC=0)
#endif
template <typename T> constexpr bool kOptimistic = STA_P3_OPTIMISTIC != 0;
// the denominator's window, on the bits: bf16 [2^-100, 2^100); fp16 [2^-5, 2^15) — one P at fp16's largest number (inf or saturated) lifts the
// sum past 2^15, and a sum of 2^-5 or more keeps the 77 keys' subnormal P (absolute error 2^-25 each) below 2^-13 of it: scores whose
// maximum lies between about -5 and +15 log2 units (-3.5 .. +10 nats) take the optimistic path, everything else the standard one
template <typename T> constexpr unsigned kDenLo = std::is_same<T, __bf16>::value ? 0x0D800000u : 0x3D000000u;
template <typename T> constexpr unsigned kDenHi = std::is_same<T, __bf16>::value ? 0x71800000u : 0x47000000u;

// One context of one head. kf holds its K operands on entry (requested a context earlier) and the NEXT context's on
// exit (`knb`, `kns`: that block's per-lane K addresses). vb / vs: this context's per-lane V^T addresses; kcb / kcs: its K
// addresses (the optimistic path's fall-back re-reads them).
// KIND 0: "" on the uncond row -> au;  1: global prompt on the cond row -> ac;  2: local prompt, ac += w (A - au).
// `sl2e`: scale * log2 e, or 1 where the scores already are in log2 units (kOptimistic).
template <typename T, int KIND>
__device__ __forceinline__ void attend3(KFr<T>& kf, const char* vb, const char* vs, const char* knb, const char* kns,
                                        const typename Tr<T>::V8& qbig, const typename Tr<T>::V4& qsm, const f32x4 kb4,
                                        const float sl2e, const float w, f32x4 (&au)[3], f32x4 (&ac)[3],
                                        const char* kcb = nullptr, const char* kcs = nullptr, const bool opt_on = true,
                                        unsigned* cnt = nullptr) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  if constexpr (kOptimistic<T>) {
    V8 vbig[3][2];
    V4 vsm[3];
    f32x4 st[NKT], o[3];
    bool need_std = true;                            // wave-uniform: the standard path runs when the optimistic one is sitting out or failed its check
    if (opt_on) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      vbig[u][0] = *(const V8*)(vb + u * 16 * VROW);
      vbig[u][1] = *(const V8*)(vb + u * 16 * VROW + 64);
      vsm[u] = *(const V4*)(vs + u * 16 * VROW);
    }
    // the k = 32 steps of all tiles first, then the k = 16 steps: four other MFMAs between a tile's two shapes (no mixed-shape hazard,
    // nothing for sta/isa_lint.py to pad)
#pragma unroll
    for (int t = 0; t < NKT; ++t) st[t] = Tr<T>::mfma(kf.big[t], qbig, (t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f});
    __builtin_amdgcn_sched_barrier(0);               // hipcc otherwise moves a tile's k = 16 step right behind its k = 32 step again
#pragma unroll
    for (int t = 0; t < NKT; ++t) st[t] = M16<T>::mfma(kf.sm[t], qsm, st[t]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[t][r] = __builtin_amdgcn_exp2f(st[t][r]);
    const V8 p0 = cat8<T>(st[0], st[1]), p1 = cat8<T>(st[2], st[3]);
    const V4 p2 = cvt4<T>(st[4]);
    __builtin_amdgcn_sched_barrier(0);
    load_k<T>(kf, knb, kns);
#pragma unroll
    for (int u = 0; u < 3; ++u) o[u] = Tr<T>::mfma(vbig[u][0], p0, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int u = 0; u < 3; ++u) o[u] = Tr<T>::mfma(vbig[u][1], p1, o[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 3; ++u) o[u] = M16<T>::mfma(vsm[u], p2, o[u]);
    // the denominator sits in lane row 2 (lanes 32..47): anything outside [2^-100, 2^100) (fp16: [2^-5, 2^15)) there -> the standard path. Tested on the
    // BITS (this file is compiled with -ffinite-math-only: a floating-point class test of inf / NaN would be folded away): one unsigned
    // compare rejects negatives, zeros, denormals, infinities and NaNs as well; the margin of 2^27 to either end of the fp32 range keeps the
    // other rows of O^T (sums of P * v) finite whenever the ones row passes.
    const bool row2 = (threadIdx.x & 48) == 32;
    const bool bad = row2 && (__float_as_uint(o[2][0]) - kDenLo<T>) >= (kDenHi<T> - kDenLo<T>);
    need_std = __builtin_amdgcn_ballot_w64(bad) != 0;
    }
    if (__builtin_expect(need_std, 0)) {
      // sta_xattn_fwd_proj_ex's statistics: a fall-back (not a sit-out) adds one to the workgroup's LDS counter — here on the rare path,
      // nothing in the common one (a lane-0 add per context evaluation there cost 2.7 % of the launch; the evaluations are counted per item)
      if (cnt && opt_on && (threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      load_k<T>(kf, kcb, kcs);                      // kf doubles as the buffer: the next context's operands are requested again below
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        vbig[u][0] = *(const V8*)(vb + u * 16 * VROW);
        vbig[u][1] = *(const V8*)(vb + u * 16 * VROW + 64);
        vsm[u] = *(const V4*)(vs + u * 16 * VROW);
      }
#pragma unroll
      for (int t = 0; t < NKT; ++t) st[t] = Tr<T>::mfma(kf.big[t], qbig, (t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int t = 0; t < NKT; ++t) st[t] = M16<T>::mfma(kf.sm[t], qsm, st[t]);
      __builtin_amdgcn_sched_barrier(0);
      load_k<T>(kf, knb, kns);
      softmax_biased(st, 1.0f, false);
      const V8 r0 = cat8<T>(st[0], st[1]), r1 = cat8<T>(st[2], st[3]);
      const V4 r2 = cvt4<T>(st[4]);
#pragma unroll
      for (int u = 0; u < 3; ++u) o[u] = Tr<T>::mfma(vbig[u][0], r0, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int u = 0; u < 3; ++u) o[u] = Tr<T>::mfma(vbig[u][1], r1, o[u]);
#pragma unroll
      for (int u = 0; u < 3; ++u) o[u] = M16<T>::mfma(vsm[u], r2, o[u]);
    }
    const float inv = bcast_row2(__builtin_amdgcn_rcpf(o[2][0]));
    const float wi = w * inv;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (KIND == 0) au[u] = o[u] * inv;
      else if (KIND == 1) ac[u] = o[u] * inv;
      else ac[u] = o[u] * wi + (ac[u] - au[u] * w);
    }
    __builtin_amdgcn_sched_barrier(0);
    return;
  }
  // V^T operands first: they land under the S^T MFMAs and the softmax
  V8 vbig[3][2];
  V4 vsm[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    vbig[u][0] = *(const V8*)(vb + u * 16 * VROW);
    vbig[u][1] = *(const V8*)(vb + u * 16 * VROW + 64);
    vsm[u] = *(const V4*)(vs + u * 16 * VROW);
  }
  f32x4 st[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    f32x4 acc = (t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f};
    acc = Tr<T>::mfma(kf.big[t], qbig, acc);
    acc = M16<T>::mfma(kf.sm[t], qsm, acc);
    st[t] = acc;
  }
  __builtin_amdgcn_sched_barrier(0);
  softmax_biased(st, sl2e, false);               // denominator: ones row of V^T
  const V8 p0 = cat8<T>(st[0], st[1]), p1 = cat8<T>(st[2], st[3]);
  const V4 p2 = cvt4<T>(st[4]);
  __builtin_amdgcn_sched_barrier(0);
  load_k<T>(kf, knb, kns);                       // the next context's K operands: requested once the scores are dead (register
                                                 // budget), they land under the PV MFMAs and the blend
  f32x4 o[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = Tr<T>::mfma(vbig[u][0], p0, acc);
    acc = Tr<T>::mfma(vbig[u][1], p1, acc);
    acc = M16<T>::mfma(vsm[u], p2, acc);
    o[u] = acc;
  }
  // the reciprocal is taken in every lane BEFORE the broadcast (a VALU instruction hipcc pads against the MFMA that wrote
  // o[2]); the asm then only moves registers
  const float inv = bcast_row2(__builtin_amdgcn_rcpf(o[2][0]));
  const float wi = w * inv;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    if (KIND == 0) au[u] = o[u] * inv;
    else if (KIND == 1) ac[u] = o[u] * inv;
    else ac[u] = o[u] * wi + (ac[u] - au[u] * w);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// Workgroup = 8 waves x 16 pixels, one HEAD PAIR, one image; walks `iters` strided pixel tiles.
//
// y rows (B operands of the projection). The launch is bound by the CU's vector-memory path, which retires about one
// 128-byte line per 3.8 cycles whatever the hit rate, loads and stores alike (profiles/r03_level0.md: the load + store skeleton
// of this kernel alone takes 141 of its 155 us). A 16x16x32 B operand wants lane (g, c) to hold 16 bytes of pixel c, so the
// natural load touches 16 rows x 64 B = 16 half-used lines per instruction. YFULL: an instruction covers 8 rows x one FULL
// line instead — lane (g, c) fetches row (c & 7), 16-byte slot g + 4 (c >> 3) of the line that holds k-steps 2m and 2m+1; two
// such loads (rows 0..7, rows 8..15) and one DPP row_ror:8 move per dword hand every lane its own pixel's slots of both
// k-steps. Same bytes and instruction count, half the lines. The ring then holds the whole row of the NEXT item.
//
// YL = 2 (query-fragment order, sta_xattn_fwd_proj_qfrag): y is not row-major at all. Its producer — the add + LayerNorm pass
// of the same block (sta_add_layernorm_qfrag), whose only consumer at this level is this kernel — writes every 16-pixel group
// as C/32 fragments of 1 KiB in B-operand lane order, so a load instruction is one fully coalesced KiB (8 whole lines, adjacent
// lanes on adjacent bytes), there is no cross-lane hand-over, and an item's 2 x NKC fragments start at the row-major byte
// offset of its first pixel. N % 16 == 0.
//
// OF (out-fragment order, with YL = 2 only): the blended output has ONE consumer as well — the to_out GEMM + residual + LayerNorm
// pass of csrc/sta_rowgemm.hip — so it leaves in that kernel's MFMA B-operand order instead of row-major: per 16-pixel group and
// batch row ten 1-KiB fragments, [2 pr] = head A's `main` registers of pair pr (lane (g, c): O^T rows 4g..4g+3 of tiles 0 | 1 of
// pixel c), [2 pr + 1] = head B's, [8 + (pr >> 1)] = the tile-2 tails of two pairs (512 bytes each: lane rows 0, 1 hold
// [A rows 32+4g.. | B rows 32+4g..]). Every store instruction is one contiguous KiB (or half of one) — no 160-byte pair
// segments at a 640-byte stride, no cross-lane exchange; sta_p3::ofrag_channel() names the channel behind every slot.
template <typename T, int NKC, int YL, bool OF>
__global__ __launch_bounds__(512, 2) void xattn_fwd_proj_p3_kernel(const P3 p) {
  static_assert(!OF || YL == 2, "out-fragment order rides on the query-fragment path (N % 16 == 0)");
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  constexpr int NWV = 8, TP = 16 * NWV;
  constexpr bool YFULL = YL == 1, YFRAG = YL == 2;
  constexpr int RING = (YFULL || YFRAG) ? NKC : 5;           // k-steps of y in flight per batch row
  constexpr unsigned YSTEP = YFRAG ? 1024u : 64u;            // bytes between consecutive k-steps of a lane's loads
  static_assert(NKC % RING == 0 && (!YFULL || NKC % 2 == 0), "k-steps per tile: a multiple of the ring depth; full-line loads pair them");
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
  const int N = p.N, C = p.C, K = p.K, W = p.W;
  const unsigned row_bytes = (unsigned)C * (unsigned)sizeof(T);
  const size_t act = (size_t)2 * N * row_bytes;
  const char* yb = (const char*)p.y + img * act;
  T* ob = (T*)((char*)p.out + img * act);
  const uint8_t* mask = p.mask + (size_t)img * N;
  const float coef_lane = p.coef[(size_t)img * K + min(lane, K > 0 ? K - 1 : 0)];
  constexpr int nwq = NT * NKC;                   // Wq fragments of the pair (1 KiB each)
  char* lds_kv = smem;                            // [ctx][hp][BLK]
  char* lds_wq = smem + kv_region(K);             // behind the blocks: the V^T over-reads of the last block stay finite

  // ---- prologue: LDS-DMA of every context (both heads: CTXB contiguous bytes each) and of the pair's Wq ----------
  // The K/V region is copied as 1-KiB pieces at 1-KiB-ALIGNED LDS offsets (a piece = one global_load_lds_dwordx4 of the wave;
  // a destination that is not 1-KiB aligned put a whole context in the wrong place on gfx950): a context is CTXB = 27.25 KiB,
  // so a piece may straddle two contexts — every lane derives its own source address from its LDS offset. Lanes past the end
  // of the region re-read its last 16 bytes into the gap in front of the Wq fragments (no exec-masked DMA).
  {
    const char* src0 = p.kv + ((size_t)img * (K + 2) * p.H + 2 * pr) * BLK;
    const unsigned cstride = (unsigned)p.H * BLK;
    const unsigned total = (unsigned)(K + 2) * CTXB;
    for (unsigned f = (unsigned)wv; f * 1024u < total; f += NWV) {
      unsigned b = f * 1024u + (unsigned)lane * 16u;
      b = b < total ? b : total - 16u;
      const unsigned c = b / (unsigned)CTXB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + (size_t)c * cstride + (b - c * CTXB)),
                                       (__attribute__((address_space(3))) void*)(lds_kv + f * 1024u), 16, 0, 0);
    }
    stage_frags(p.wq + (size_t)pr * nwq * FRAG, lds_wq, nwq, wv, NWV, lane);
  }
  // Work items = (pixel tile of the workgroup, 16-pixel sub-tile): a wave takes the next item from an LDS counter when it is one
  // item ahead of its y ring, instead of owning sub-tile `wv` of every tile: the second-dispatched half of the workgroup loses
  // the issue arbitration on every SIMD (its waves needed 25 % longer for the same 16 sub-tiles: profiles/r03_level0.md) and
  // disc-crossing sub-tiles cost two more contexts — with the queue every wave ends within one item of the others.
  const int mine = (p.tiles - wt + W - 1) / W;
  const int iters = mine < p.iters ? mine : p.iters;
  const int nitems = iters * NWV;
  unsigned* qcount = (unsigned*)(lds_wq + (size_t)nwq * FRAG);     // behind the Wq fragments: [0] the item queue, [1], [2] this workgroup's
                                                                     // optimistic-softmax attempts / fall-backs (statistics)
  if (threadIdx.x == 0) *qcount = 2u * NWV;                          // items 0 .. 2 NWV - 1 are handed out statically below
  const __amdgpu_buffer_rsrc_t y_srd = make_srd(yb, (unsigned)act);
  const __amdgpu_buffer_rsrc_t o_srd = make_srd(ob, (unsigned)act);
  const unsigned row1 = (unsigned)N * row_bytes;
  auto px0_of = [&](int q) -> int { return (wt + (q >> 3) * W) * TP + (q & (NWV - 1)) * 16; };
  // per-lane byte offset of this lane's y load(s) for item `q` (pixels >= N and items past the end are pushed out of the
  // descriptor's range: they read as zeros). Half-line shape: row c16, 16-byte slot g of a k-step. Full-line shape: two loads
  // per line pair — `half` 0: rows 0..7 of the item's 16 pixels, 1: rows 8..15 — lane (g, c) takes row (c & 7), slot g + 4 (c >> 3)
  auto voff_of = [&](int q, int half = 0) -> unsigned {
    if constexpr (YFRAG) {      // fragment s of the item's group: byte offset of its first pixel's row + 1024 s + 16 lane
      const int px0 = px0_of(q);
      return (q < nitems && px0 < N) ? (unsigned)px0 * row_bytes + (unsigned)lane * 16u : 0xfffffff0u;
    }
    const int px = px0_of(q) + (YFULL ? (c16 & 7) + 8 * half : c16);
    const unsigned slot = YFULL ? (unsigned)(g + 4 * (c16 >> 3)) : (unsigned)g;
    return (q < nitems && px < N) ? (unsigned)px * row_bytes + slot * 16u : 0xfffffff0u;
  };
  auto mask_of = [&](int q) -> unsigned {
    const int px = px0_of(q) + c16;
    return mask[(q < nitems && px < N) ? px : 0];
  };
  int qcur = wv, qnext = NWV + wv;
  // ring slot j: half-line shape = k-step j (mod RING) of both batch rows; full-line shape = (line pair j >> 1, rows-half j & 1)
  V8 yr0[RING], yr1[RING];
  unsigned voff = voff_of(qcur), voffn = voff_of(qnext);
  unsigned voffh = YFULL ? voff_of(qcur, 1) : 0u, voffhn = YFULL ? voff_of(qnext, 1) : 0u;
  unsigned mb = mask_of(qcur);
#pragma unroll
  for (int j = 0; j < RING; ++j) {
    const unsigned vo = (YFULL && (j & 1)) ? voffh : voff;
    const unsigned so = YFULL ? 128u * (unsigned)(j >> 1) : YSTEP * (unsigned)j;
    yr0[j] = srd_load16<V8>(y_srd, vo, so);
    yr1[j] = srd_load16<V8>(y_srd, vo, row1 + so);
  }
  const f32x4 kb4 = last_tile_bias(g, p.M);
  const float sl2e = p.sl2e;
  const unsigned kmask = (1u << K) - 1u;
  // per-lane byte offsets into a (ctx, head) block
  // (slot of lane row g inside row c16 of a 16-row tile: sta_p3::swz_big / swz_small — bank-conflict-free operand reads)
  const int koffb = c16 * KROW + 16 * swz_big(g, c16), koffs = c16 * KROW + 64 + 8 * swz_small(g, c16);
  const int voffb = KBYTES + c16 * VROW + 16 * swz_big(g, c16), voffs = KBYTES + c16 * VROW + 128 + 8 * swz_small(g, c16);
  const V8* wf = (const V8*)lds_wq + lane;
  // optimistic softmax or not: the caller's state word (launches still to sit out; sta_xattn_fwd_proj_ex) — one value per launch
#ifndef STA_P3_STATS
#define STA_P3_STATS 1        // 0: no statistics / switch code in the kernel (same-box A/B builds: tools/lib_ab.py ... nostats=-DSTA_P3_STATS=0)
#endif
#if STA_P3_STATS
  const bool opt_on = p.stats == nullptr || __builtin_amdgcn_readfirstlane((int)__builtin_nontemporal_load(p.stats)) == 0;
#else
  constexpr bool opt_on = true;
#endif
  if (threadIdx.x == 0) { qcount[1] = 0u; qcount[2] = 0u; }
  wait_dma_and_sync();
  if constexpr (kOptimistic<T>) {
    // scores in log2 units: scale * log2 e goes into the pair's Wq fragments, once per workgroup, in LDS (nwq fragments of 64 lanes x 8
    // values; W' = round16(W * sl2e)) — the projection then delivers q * scale * log2 e with no per-item multiply
    for (int f = (int)threadIdx.x; f < nwq * 64; f += 64 * NWV) {
      V8 w = ((const V8*)lds_wq)[f];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = (T)((float)w[j] * sl2e);
      ((V8*)lds_wq)[f] = w;
    }
    __syncthreads();
  }

  unsigned* const cnt_a = (STA_P3_STATS && p.stats) ? qcount + 1 : nullptr;      // LDS counters; every 8th workgroup reports them (tail)
  while (qcur < nitems) {
    // ---- projection: 5 column tiles x both batch rows; Wq fragments one k-step ahead -------------------------------
    f32x4 qa0[NT], qa1[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      qa0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      qa1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    V8 a[2][NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) a[0][u] = wf[u * 64];
#pragma unroll
    for (int s = 0; s < NKC; ++s) {
      if (s + 1 < NKC) {
#pragma unroll
        for (int u = 0; u < NT; ++u) a[(s + 1) & 1][u] = wf[((s + 1) * NT + u) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (YFULL) {
        // slot s & ~1 (rows 0..7): lanes c < 8 hold their own pixel's step-2m slots, lanes c >= 8 pixel c-8's step-2m+1 slots;
        // slot s | 1 (rows 8..15): lanes c < 8 hold pixel c+8's step-2m slots, lanes c >= 8 their own step-2m+1 slots.
        // even step: c < 8 keeps A, c >= 8 takes ror8(B);  odd step: c < 8 takes ror8(A), c >= 8 keeps B.
        const int ja = s & ~1, jb = s | 1;
        const u32x4 A0 = __builtin_bit_cast(u32x4, yr0[ja]), B0 = __builtin_bit_cast(u32x4, yr0[jb]);
        const u32x4 A1 = __builtin_bit_cast(u32x4, yr1[ja]), B1 = __builtin_bit_cast(u32x4, yr1[jb]);
        u32x4 r0, r1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if ((s & 1) == 0) {
            r0[q] = (unsigned)__builtin_amdgcn_update_dpp((int)A0[q], (int)B0[q], 0x128, 0xF, 0xC, false);
            r1[q] = (unsigned)__builtin_amdgcn_update_dpp((int)A1[q], (int)B1[q], 0x128, 0xF, 0xC, false);
          } else {
            r0[q] = (unsigned)__builtin_amdgcn_update_dpp((int)B0[q], (int)A0[q], 0x128, 0xF, 0x3, false);
            r1[q] = (unsigned)__builtin_amdgcn_update_dpp((int)B1[q], (int)A1[q], 0x128, 0xF, 0x3, false);
          }
        }
        const V8 b0 = __builtin_bit_cast(V8, r0), b1 = __builtin_bit_cast(V8, r1);
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          qa0[u] = Tr<T>::mfma(a[s & 1][u], b0, qa0[u]);
          qa1[u] = Tr<T>::mfma(a[s & 1][u], b1, qa1[u]);
        }
        if (s & 1) {       // the line pair is consumed: request the same pair of the NEXT item into both slots
          const unsigned so = 128u * (unsigned)(s >> 1);
          yr0[ja] = srd_load16<V8>(y_srd, voffn, so);
          yr1[ja] = srd_load16<V8>(y_srd, voffn, row1 + so);
          yr0[jb] = srd_load16<V8>(y_srd, voffhn, so);
          yr1[jb] = srd_load16<V8>(y_srd, voffhn, row1 + so);
        }
      } else {
        const int j = s % RING;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          qa0[u] = Tr<T>::mfma(a[s & 1][u], yr0[j], qa0[u]);
          qa1[u] = Tr<T>::mfma(a[s & 1][u], yr1[j], qa1[u]);
        }
        {
          // refill this ring slot with k-step s + RING: of this item, or of the next one (zeros past the last item)
          const bool wrap = s + RING >= NKC;              // compile time after unrolling
          const unsigned vo = wrap ? voffn : voff;
          const unsigned so = YSTEP * (unsigned)(wrap ? s + RING - NKC : s + RING);
          yr0[j] = srd_load16<V8>(y_srd, vo, so);
          yr1[j] = srd_load16<V8>(y_srd, vo, row1 + so);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // K operands of the first context (head A, ctx 0): requested here, they land under the accumulator conversions
    KFr<T> kf;
    load_k<T>(kf, lds_kv + koffb, lds_kv + koffs);
    const int px_own = px0_of(qcur) + c16;
    const bool valid = px_own < N;
    // the item after next: one LDS atomic by lane 0, consumed at the end of this item (its latency sits under the attention)
    const unsigned mbn = mask_of(qnext);
    const unsigned mbits = valid ? (mb & kmask) : 0u;
    // local contexts some pixel of this wave needs (wave-uniform bit set)
    unsigned wneed = 0;
    for (int i = 0; i < K; ++i) wneed |= __ballot((mbits >> i) & 1u) ? (1u << i) : 0u;
    unsigned qtake = 0;
    if (lane == 0) {
      qtake = atomicAdd(qcount, 1u);
      // statistics: this item's optimistic context evaluations (both heads), no return value, in the same masked region as the queue
      if (STA_P3_STATS) __hip_atomic_fetch_add(qcount + 1, opt_on ? 2u * (2u + (unsigned)__builtin_popcount(wneed)) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    // accumulators -> S^T B operands (rounded to T once). Head A: tiles 0 | 1 (+ tile 2 rows g < 2), head B: tiles 3 | 4
    // (+ tile 2 rows g >= 2); the small operand (tile 2) serves both heads, the K images carry the zeros.
    const float sm_scale = kOptimistic<T> ? 1.0f : sl2e;
    const V8 qA0 = cat8<T>(qa0[0], qa0[1]), qA1 = cat8<T>(qa1[0], qa1[1]);
    const V8 qB0 = cat8<T>(qa0[3], qa0[4]), qB1 = cat8<T>(qa1[3], qa1[4]);
    const V4 qs0 = cvt4<T>(qa0[2]), qs1 = cvt4<T>(qa1[2]);

    // ---- attention + blend, head A then head B ------------------------------------------------------------------
    // `seg`: byte offset of this lane's pixel row, pair segment (160 bytes at 160 pr), uncond row; + row1 = cond row
    const unsigned seg = (unsigned)px_own * row_bytes + (unsigned)(2 * pr * D) * (unsigned)sizeof(T);
    const unsigned seg1 = seg + row1;
    auto head = [&](auto hb_tag, const V8& q0, const V8& q1, OutRow& ou, OutRow& oc) {
      constexpr int HB = decltype(hb_tag)::value;
      f32x4 au[3], ac[3];
      const char* blk = lds_kv + HB * BLK;                       // (ctx 0, this head); contexts are CTXB apart
      const char* other = lds_kv + (HB ^ 1) * BLK;               // what follows this head: head B's ctx 0, or head A's of the next item
      // block whose K operands to request during context `cur` (0, 1, or 2 + i): the next needed local context, else `other`
      auto next_of = [&](int first_local) -> const char* {
        const unsigned rest = wneed >> first_local;
        return rest ? blk + (size_t)(2 + first_local + __builtin_ctz(rest)) * CTXB : other;
      };
      attend3<T, 0>(kf, blk + voffb, blk + voffs, blk + CTXB + koffb, blk + CTXB + koffs, q0, qs0, kb4, sm_scale, 0.f, au, ac, blk + koffb, blk + koffs, opt_on, cnt_a);
      {
        const char* nx = next_of(0);
        attend3<T, 1>(kf, blk + CTXB + voffb, blk + CTXB + voffs, nx + koffb, nx + koffs, q1, qs1, kb4, sm_scale, 0.f, au, ac, blk + CTXB + koffb, blk + CTXB + koffs, opt_on, cnt_a);
      }
      for (int i = 0; i < K; ++i) {
        if (!((wneed >> i) & 1u)) continue;
        const float cw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef_lane), i));
        const float w = ((mbits >> i) & 1u) ? cw : 0.f;
        const char* cb = blk + (size_t)(2 + i) * CTXB;
        const char* nx = next_of(i + 1);
        attend3<T, 2>(kf, cb + voffb, cb + voffs, nx + koffb, nx + koffs, q1, qs1, kb4, sm_scale, w, au, ac, cb + koffb, cb + koffs, opt_on, cnt_a);
      }
      ou = pack_out<T>(au);
      oc = pack_out<T>(ac);
    };
    OutRow au_A, ac_A, au_B, ac_B;
    head(std::integral_constant<int, 0>{}, qA0, qA1, au_A, ac_A);
    if constexpr (OF) {
      // out-fragment order: fragment 2 pr of the item's group, one contiguous KiB per batch row
      const unsigned fb = valid ? (unsigned)px0_of(qcur) * row_bytes + (unsigned)lane * 16u + 2048u * (unsigned)pr : 0xfffffff0u;
      __builtin_amdgcn_raw_buffer_store_b128(au_A.main, o_srd, fb, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(ac_A.main, o_srd, fb, row1, 0);
      head(std::integral_constant<int, 1>{}, qB0, qB1, au_B, ac_B);
      __builtin_amdgcn_raw_buffer_store_b128(au_B.main, o_srd, fb, 1024u, 0);
      __builtin_amdgcn_raw_buffer_store_b128(ac_B.main, o_srd, fb, row1 + 1024u, 0);
      // tails: lane rows 0, 1 hold [A rows 32 + 4g .. | B rows 32 + 4g ..]; two pairs share fragment 8 + (pr >> 1)
      const u32x4 zu = {au_A.tail[0], au_A.tail[1], au_B.tail[0], au_B.tail[1]}, zc = {ac_A.tail[0], ac_A.tail[1], ac_B.tail[0], ac_B.tail[1]};
      const unsigned zb = (valid && g < 2) ? (unsigned)px0_of(qcur) * row_bytes + (unsigned)lane * 16u + 8192u + 1024u * (unsigned)(pr >> 1) + 512u * (unsigned)(pr & 1)
                                           : 0xfffffff0u;
      __builtin_amdgcn_raw_buffer_store_b128(zu, o_srd, zb, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(zc, o_srd, zb, row1, 0);
    } else {
      // head A's dims 0..31 of both batch rows: bytes 0..63 of the pair segment, 16 bytes per lane, no cross-lane exchange
      __builtin_amdgcn_raw_buffer_store_b128(au_A.main, o_srd, valid ? seg + 16u * (unsigned)g : 0xfffffff0u, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(ac_A.main, o_srd, valid ? seg1 + 16u * (unsigned)g : 0xfffffff0u, 0, 0);
      head(std::integral_constant<int, 1>{}, qB0, qB1, au_B, ac_B);
      store_pair_rest(o_srd, seg, valid, g, au_A.tail, au_B);
      store_pair_rest(o_srd, seg1, valid, g, ac_A.tail, ac_B);
    }
    mb = mbn;
    qcur = qnext;
    qnext = (int)__builtin_amdgcn_readfirstlane(qtake);
    voff = voffn;
    voffn = voff_of(qnext);
    if constexpr (YFULL) {
      voffh = voffhn;
      voffhn = voff_of(qnext, 1);
    }
  }
  // Statistics: every 8th workgroup of the launch is a SAMPLE (all 2048 workgroups adding to three words cost 6 us of a 235 us launch:
  // same-address atomics serialise in L2). Per sampled workgroup: LDS counters (attend3) -> one 64-bit atomic {evaluations, fall-backs}
  // + one for the finished count.
  if (STA_P3_STATS && p.stats && (lin & 7) == 0) {
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long* cnt = (unsigned long long*)(p.stats + 2);
      if (qcount[1]) atomicAdd(cnt, (unsigned long long)qcount[1] | ((unsigned long long)qcount[2] << 32));
      __threadfence();
      // The LAST sampled workgroup to get here folds the launch's counts into the totals and decides whether the next launches sit the
      // optimistic softmax out — more than an eighth of the sampled wave-level context evaluations fell back: the next 64 launches run
      // the standard softmax only (every context through what is otherwise the fall-back: exact, ~4 % slower than the optimistic path
      // on friendly logits instead of both paths per context on hostile ones). Every workgroup has read word 0 when it started, and
      // the last sample cannot finish before... it may: workgroups that have not started yet then see the NEW word 0 — either value is a
      // valid choice per workgroup (both paths are exact), so the launch stays correct.
      const unsigned nsamp = (gridDim.x * gridDim.y + 7u) >> 3;
      if (atomicAdd(p.stats + 1, 1u) == nsamp - 1u) {
        __threadfence();
        unsigned* s = p.stats;
        const unsigned long long c = atomicAdd(cnt, 0ull);
        const unsigned att = (unsigned)c, fb = (unsigned)(c >> 32);
        s[4] += att; s[5] += fb; s[6] += 1u;
        if (s[0] > 0u) { s[0] -= 1u; s[7] += 1u; }
        else if (fb * 8u > att) s[0] = 64u;
        s[1] = 0u; s[2] = 0u; s[3] = 0u;
        __threadfence();
      }
    }
  }
}

template <typename T, int NKC, int YL, bool OF = false>
int launch_p3(P3 p, int n_img, hipStream_t st) {
  constexpr int TP = 128;
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
  if (!attr.ensure((const void*)xattn_fwd_proj_p3_kernel<T, NKC, YL, OF>, 160 * 1024))
    return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(fwd proj p3) failed");
  hipLaunchKernelGGL((xattn_fwd_proj_p3_kernel<T, NKC, YL, OF>), dim3(p.W * pairs, n_img), dim3(512), lds, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "fwd proj p3 launch: %s", hipGetErrorString(e));
}

}  // namespace

namespace sta_p3 {

int pack_kv(const void* k, const void* v, void* packed, int n_ctx, int M, int C, int heads, int dtype, hipStream_t st) {
  const dim3 grid(n_ctx * heads);
  if (dtype == STA_BF16)
    hipLaunchKernelGGL(pack_kv_p3_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)k, (const __bf16*)v, (__bf16*)packed, M, C, heads);
  else
    hipLaunchKernelGGL(pack_kv_p3_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16*)k, (const _Float16*)v, (_Float16*)packed, M, C, heads);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "pack_kv_p3 launch: %s", hipGetErrorString(e));
}

int forward(const void* y, const void* wq_pair, const void* kv, const uint8_t* mask, const float* coef, void* out, int n_img,
            int N, int C, int heads, int M, int K, float sl2e, int dtype, hipStream_t st, bool qfrag, bool ofrag, unsigned* stats) {
  P3 p{};
  p.stats = stats;
  p.y = y; p.wq = (const char*)wq_pair; p.kv = (const char*)kv; p.mask = mask; p.coef = coef; p.out = out;
  p.N = N; p.C = C; p.H = heads; p.M = M; p.K = K; p.sl2e = sl2e;
  if (ofrag && !(qfrag && C == 320)) return sta_fail(STA_E_UNSUP, "out-fragment order needs y in query-fragment order and C = 320");
  if (qfrag) {      // y in query-fragment order (sta_add_layernorm_qfrag): 1-KiB coalesced loads, no hand-over
    if (N % 16) return sta_fail(STA_E_UNSUP, "query-fragment order needs N %% 16 == 0 (N=%d)", N);
    if (ofrag) {
      return dtype == STA_BF16 ? launch_p3<__bf16, 10, 2, true>(p, n_img, st) : launch_p3<_Float16, 10, 2, true>(p, n_img, st);
    }
    if (C == 320) return dtype == STA_BF16 ? launch_p3<__bf16, 10, 2>(p, n_img, st) : launch_p3<_Float16, 10, 2>(p, n_img, st);
    return dtype == STA_BF16 ? launch_p3<__bf16, 5, 2>(p, n_img, st) : launch_p3<_Float16, 5, 2>(p, n_img, st);
  }
  // row-major y. Full-line loads pair the k-steps: C = 320 (10 steps) takes them, C = 160 (5 steps) keeps the half-line shape
  if (C == 320) return dtype == STA_BF16 ? launch_p3<__bf16, 10, 1>(p, n_img, st) : launch_p3<_Float16, 10, 1>(p, n_img, st);
  return dtype == STA_BF16 ? launch_p3<__bf16, 5, 0>(p, n_img, st) : launch_p3<_Float16, 5, 0>(p, n_img, st);
}

}  // namespace sta_p3
