// p3_chain_probe.hip — does a SECOND softmax chain per wave buy the level-0 head-pair kernel its arithmetic half?
// ("put a second (context, head) chain in flight per wave"; DESIGN section 8 next (2): a 4-wave x 32-pixel
// geometry, one wave per SIMD with 512 registers.)
//
// A timing probe of the ATTENTION PHASE only, built from the product's own device code (this file #includes
// csrc/sta_xattn_proj3.hip: attend3, load_k, bcast_row2, the operand images — nothing is restated): every wave keeps the q
// operands of its pixels in registers and runs `reps` times what an item without local contexts runs — head A: ""-context
// on the uncond row, global prompt on the cond row; head B the same — against K / V^T images resident in LDS; no projection, no y
// loads, no stores inside the loop (a 16-byte xor-checksum per wave at the end keeps the work alive).
//   NP = 1, 8 waves: the product's geometry (two waves per SIMD, one 16-pixel group per wave)
//   NP = 1, 4 waves: one wave per SIMD, one group (what the second wave of a SIMD contributes)
//   NP = 2, 4 waves: one wave per SIMD, TWO 16-pixel groups per wave — every K / V^T operand read serves two MFMAs, the two
//                    groups' softmax chains are independent instruction streams the compiler may interleave
// Same pixels x contexts per launch in the first and the third. Standalone build (tools/p3_chain_probe.py): hipcc -shared.
#include "../../diffusion-spacetime-attn_amd/csrc/sta_xattn_proj3.hip"

// what csrc/sta_xattn.hip defines for the library
thread_local char g_sta_err[256] = "";
StaOpt g_sta_opt[STA_OPT_COUNT];
int sta_fail(int code, const char*, ...) { return code; }

namespace {

template <typename T, int KIND, int NP>
__device__ __forceinline__ void attend3n(KFr<T>& kf, const char* vb, const char* vs, const char* knb, const char* kns,
                                         const typename Tr<T>::V8 (&qbig)[NP], const typename Tr<T>::V4 (&qsm)[NP], const f32x4 kb4,
                                         const float sl2e, const float (&w)[NP], f32x4 (&au)[NP][3], f32x4 (&ac)[NP][3]) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  V8 vbig[3][2];
  V4 vsm[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    vbig[u][0] = *(const V8*)(vb + u * 16 * VROW);
    vbig[u][1] = *(const V8*)(vb + u * 16 * VROW + 64);
    vsm[u] = *(const V4*)(vs + u * 16 * VROW);
  }
  f32x4 st[NP][NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      f32x4 acc = (t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f};
      acc = Tr<T>::mfma(kf.big[t], qbig[q], acc);
      st[q][t] = acc;
    }
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int q = 0; q < NP; ++q) st[q][t] = M16<T>::mfma(kf.sm[t], qsm[q], st[q][t]);
  __builtin_amdgcn_sched_barrier(0);
  V8 p0[NP], p1[NP];
  V4 p2[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    softmax_biased(st[q], sl2e, false);
    p0[q] = cat8<T>(st[q][0], st[q][1]);
    p1[q] = cat8<T>(st[q][2], st[q][3]);
    p2[q] = cvt4<T>(st[q][4]);
  }
  __builtin_amdgcn_sched_barrier(0);
  load_k<T>(kf, knb, kns);
  f32x4 o[NP][3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = Tr<T>::mfma(vbig[u][0], p0[q], acc);
      o[q][u] = acc;
    }
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int q = 0; q < NP; ++q) o[q][u] = Tr<T>::mfma(vbig[u][1], p1[q], o[q][u]);
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int q = 0; q < NP; ++q) o[q][u] = M16<T>::mfma(vsm[u], p2[q], o[q][u]);
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const float inv = bcast_row2(__builtin_amdgcn_rcpf(o[q][2][0]));
    const float wi = w[q] * inv;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (KIND == 0) au[q][u] = o[q][u] * inv;
      else if (KIND == 1) ac[q][u] = o[q][u] * inv;
      else ac[q][u] = o[q][u] * wi + (ac[q][u] - au[q][u] * w[q]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// Both mandatory contexts of a head as TWO INTERLEAVED chains of one wave (they are independent: other q row, other K / V^T, other
// output): the S^T MFMAs of context 1 are issued with the softmax of context 0 between them, the PV MFMAs of context 0 with the
// softmax of context 1, the PV MFMAs of context 1 with the blend of context 0 — sched_group_barrier pins "one MFMA, then IL vector
// instructions". Costs the second context's K operands (30 registers) and scores (20) live beside the first's.
// kf holds K(ctx 0) on entry and K of the block behind `knb` / `kns` on exit.
template <typename T, int IL>
__device__ __forceinline__ void attend_pair_il(KFr<T>& kf, const char* blk, const int koffb, const int koffs, const int voffb, const int voffs,
                                               const char* knb, const char* kns, const typename Tr<T>::V8& q0, const typename Tr<T>::V4& qs0,
                                               const typename Tr<T>::V8& q1, const typename Tr<T>::V4& qs1, const f32x4 kb4, const float sl2e,
                                               f32x4 (&au)[3], f32x4 (&ac)[3]) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  constexpr int M_MFMA = 0x8, M_VALU = 0x402;
  KFr<T> kf1;
  load_k<T>(kf1, blk + CTXB + koffb, blk + CTXB + koffs);
  V8 v0b[3][2], v1b[3][2];
  V4 v0s[3], v1s[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    v0b[u][0] = *(const V8*)(blk + voffb + u * 16 * VROW);
    v0b[u][1] = *(const V8*)(blk + voffb + u * 16 * VROW + 64);
    v0s[u] = *(const V4*)(blk + voffs + u * 16 * VROW);
  }
  f32x4 st0[NKT], st1[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    f32x4 acc = (t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f};
    acc = Tr<T>::mfma(kf.big[t], q0, acc);
    st0[t] = M16<T>::mfma(kf.sm[t], qs0, acc);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- S^T of context 1 with the softmax of context 0 between its MFMAs
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    f32x4 acc = (t == NKT - 1) ? kb4 : f32x4{0.f, 0.f, 0.f, 0.f};
    st1[t] = Tr<T>::mfma(kf1.big[t], q1, acc);
  }
#pragma unroll
  for (int t = 0; t < NKT; ++t) st1[t] = M16<T>::mfma(kf1.sm[t], qs1, st1[t]);
  softmax_biased(st0, sl2e, false);
  const V8 p00 = cat8<T>(st0[0], st0[1]), p01 = cat8<T>(st0[2], st0[3]);
  const V4 p02 = cvt4<T>(st0[4]);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    __builtin_amdgcn_sched_group_barrier(M_MFMA, 1, 0);
    __builtin_amdgcn_sched_group_barrier(M_VALU, IL, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    v1b[u][0] = *(const V8*)(blk + CTXB + voffb + u * 16 * VROW);
    v1b[u][1] = *(const V8*)(blk + CTXB + voffb + u * 16 * VROW + 64);
    v1s[u] = *(const V4*)(blk + CTXB + voffs + u * 16 * VROW);
  }
  // ---- PV of context 0 with the softmax of context 1 between its MFMAs
  f32x4 o0[3], o1[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) o0[u] = Tr<T>::mfma(v0b[u][0], p00, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
  for (int u = 0; u < 3; ++u) o0[u] = Tr<T>::mfma(v0b[u][1], p01, o0[u]);
#pragma unroll
  for (int u = 0; u < 3; ++u) o0[u] = M16<T>::mfma(v0s[u], p02, o0[u]);
  softmax_biased(st1, sl2e, false);
  const V8 p10 = cat8<T>(st1[0], st1[1]), p11 = cat8<T>(st1[2], st1[3]);
  const V4 p12 = cvt4<T>(st1[4]);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    __builtin_amdgcn_sched_group_barrier(M_MFMA, 1, 0);
    __builtin_amdgcn_sched_group_barrier(M_VALU, IL + 1, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  load_k<T>(kf, knb, kns);
  // ---- PV of context 1 with the blend of context 0 between its MFMAs
#pragma unroll
  for (int u = 0; u < 3; ++u) o1[u] = Tr<T>::mfma(v1b[u][0], p10, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
  for (int u = 0; u < 3; ++u) o1[u] = Tr<T>::mfma(v1b[u][1], p11, o1[u]);
#pragma unroll
  for (int u = 0; u < 3; ++u) o1[u] = M16<T>::mfma(v1s[u], p12, o1[u]);
  {
    const float inv = bcast_row2(__builtin_amdgcn_rcpf(o0[2][0]));
#pragma unroll
    for (int u = 0; u < 3; ++u) au[u] = o0[u] * inv;
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    const float inv = bcast_row2(__builtin_amdgcn_rcpf(o1[2][0]));
#pragma unroll
    for (int u = 0; u < 3; ++u) ac[u] = o1[u] * inv;
  }
  __builtin_amdgcn_sched_barrier(0);
}

// The optimistic softmax: scores arrive in log2 units (the caller folded scale * log2 e into q), P = exp2(S) WITHOUT the running maximum —
// no v_max3 chain, no row butterfly, no scale-and-subtract FMAs. Exact whenever the denominator stays a normal fp32 number, which the
// 8-bit exponent of bf16 P operands allows for |logit| < ~80; GUARD: the denominator's class is checked (zero / denormal / inf / NaN) and a
// wave-uniform branch would take the standard path (here: only the check and a dummy branch, for timing).
template <typename T, bool GUARD>
__device__ __forceinline__ void attend3_nomax(KFr<T>& kf, const char* vb, const char* vs, const char* knb, const char* kns,
                                              const typename Tr<T>::V8& qbig, const typename Tr<T>::V4& qsm, const f32x4 kb4, f32x4 (&a)[3], unsigned* flag) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
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
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) st[t][r] = __builtin_amdgcn_exp2f(st[t][r]);
  const V8 p0 = cat8<T>(st[0], st[1]), p1 = cat8<T>(st[2], st[3]);
  const V4 p2 = cvt4<T>(st[4]);
  __builtin_amdgcn_sched_barrier(0);
  load_k<T>(kf, knb, kns);
  f32x4 o[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = Tr<T>::mfma(vbig[u][0], p0, acc);
    acc = Tr<T>::mfma(vbig[u][1], p1, acc);
    acc = M16<T>::mfma(vsm[u], p2, acc);
    o[u] = acc;
  }
  if constexpr (GUARD) {
    // lane row 2 holds the denominator: anything but a normal positive number sends the WAVE down the standard path
    const bool bad = __builtin_isfpclass(o[2][0], 0x3ff & ~0x100) && (threadIdx.x & 63) >= 32 && (threadIdx.x & 63) < 48;
    if (__builtin_amdgcn_ballot_w64(bad)) atomicAdd(flag, 1u);
  }
  const float inv = bcast_row2(__builtin_amdgcn_rcpf(o[2][0]));
#pragma unroll
  for (int u = 0; u < 3; ++u) a[u] = o[u] * inv;
  __builtin_amdgcn_sched_barrier(0);
}

// MODE 0: the product's attend3 (NP = 1 only); 1: attend3n (MFMA clusters grouped across the NP pixel groups); 4 / 5: attend3_nomax without / with guard;
// 2 / 3: attend_pair_il with 6 / 4 vector instructions behind every MFMA (NP = 1)
template <typename T, int NP, int NWV, int MODE>
__global__ __launch_bounds__(64 * NWV, NWV == 8 ? 2 : 1) void chain_probe_kernel(const char* kvimg, const T* qsrc, unsigned* sink, int reps, float sl2e) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  char* lds_kv = smem;
  // two contexts x two heads = 2 CTXB bytes, copied as 1-KiB pieces (as the product's prologue does)
  const unsigned total = 2u * CTXB;
  for (unsigned f = (unsigned)wv; f * 1024u < total; f += NWV) {
    unsigned b = f * 1024u + (unsigned)lane * 16u;
    b = b < total ? b : total - 16u;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kvimg + (size_t)blockIdx.x % 4 * total + b),
                                     (__attribute__((address_space(3))) void*)(lds_kv + f * 1024u), 16, 0, 0);
  }
  V8 qA0[NP], qA1[NP], qB0[NP], qB1[NP];
  V4 qs0[NP], qs1[NP];
  const T* qw = qsrc + ((size_t)(blockIdx.x * NWV + wv) * NP) * 64 * 40;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const T* ql = qw + (size_t)q * 64 * 40 + lane * 40;
    qA0[q] = *(const V8*)(ql);
    qA1[q] = *(const V8*)(ql + 8);
    qB0[q] = *(const V8*)(ql + 16);
    qB1[q] = *(const V8*)(ql + 24);
    qs0[q] = *(const V4*)(ql + 32);
    qs1[q] = *(const V4*)(ql + 36);
  }
  const f32x4 kb4 = last_tile_bias(g, 77);
  const int koffb = c16 * KROW + 16 * g, koffs = c16 * KROW + 64 + 8 * g;
  const int voffb = KBYTES + c16 * VROW + 16 * g, voffs = KBYTES + c16 * VROW + 128 + 8 * g;
  wait_dma_and_sync();
  KFr<T> kf;
  load_k<T>(kf, lds_kv + koffb, lds_kv + koffs);
  u32x4 chk = {0u, 0u, 0u, 0u};
  float w[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) w[q] = 0.f;
  auto head = [&](auto hb_tag, const V8 (&q0)[NP], const V8 (&q1)[NP]) __attribute__((always_inline)) {
    constexpr int HB = decltype(hb_tag)::value;
    const char* blk = lds_kv + HB * BLK;
    const char* other = lds_kv + (HB ^ 1) * BLK;
    f32x4 au[NP][3], ac[NP][3];
    if constexpr (MODE == 0) {
      static_assert(MODE != 0 || NP == 1, "the product's attend3 takes one pixel group");
      attend3<T, 0>(kf, blk + voffb, blk + voffs, blk + CTXB + koffb, blk + CTXB + koffs, q0[0], qs0[0], kb4, sl2e, 0.f, au[0], ac[0]);
      attend3<T, 1>(kf, blk + CTXB + voffb, blk + CTXB + voffs, other + koffb, other + koffs, q1[0], qs1[0], kb4, sl2e, 0.f, au[0], ac[0]);
    } else if constexpr (MODE >= 4) {
      attend3_nomax<T, MODE == 5>(kf, blk + voffb, blk + voffs, blk + CTXB + koffb, blk + CTXB + koffs, q0[0], qs0[0], kb4, au[0], sink);
      attend3_nomax<T, MODE == 5>(kf, blk + CTXB + voffb, blk + CTXB + voffs, other + koffb, other + koffs, q1[0], qs1[0], kb4, ac[0], sink);
    } else if constexpr (MODE >= 2) {
      attend_pair_il<T, MODE == 2 ? 6 : 4>(kf, blk, koffb, koffs, voffb, voffs, other + koffb, other + koffs, q0[0], qs0[0], q1[0], qs1[0], kb4, sl2e, au[0], ac[0]);
    } else {
      attend3n<T, 0, NP>(kf, blk + voffb, blk + voffs, blk + CTXB + koffb, blk + CTXB + koffs, q0, qs0, kb4, sl2e, w, au, ac);
      attend3n<T, 1, NP>(kf, blk + CTXB + voffb, blk + CTXB + voffs, other + koffb, other + koffs, q1, qs1, kb4, sl2e, w, au, ac);
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const OutRow ou = pack_out<T>(au[q]), oc = pack_out<T>(ac[q]);
      chk = chk ^ ou.main ^ oc.main;
      chk[0] ^= ou.tail[0] ^ oc.tail[1];
    }
  };
  for (int r = 0; r < reps; ++r) {
    head(std::integral_constant<int, 0>{}, qA0, qA1);
    head(std::integral_constant<int, 1>{}, qB0, qB1);
  }
  *(u32x4*)(sink + ((size_t)(blockIdx.x * NWV + wv) * 64 + lane) * 4) = chk;
}

template <typename T, int NP, int NWV, int MODE>
int launch_probe(const void* kv, const void* q, void* sink, int reps, float sl2e, hipStream_t st) {
  const int lds = 150 * 1024;         // one workgroup per CU, as in the product
  static StaLdsAttr attr;
  if (!attr.ensure((const void*)chain_probe_kernel<T, NP, NWV, MODE>, 160 * 1024)) return -3;
  hipLaunchKernelGGL((chain_probe_kernel<T, NP, NWV, MODE>), dim3(256), dim3(64 * NWV), lds, st, (const char*)kv, (const T*)q, (unsigned*)sink, reps, sl2e);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

// variant: 0 = NP 1 x 8 waves, product attend3;  1 = NP 1 x 4 waves, product attend3;  2 = NP 2 x 4 waves, grouped;  3 = NP 1 x 8 waves, grouped code path
//          4 = NP 2 x 8 waves (two waves per SIMD, two groups each: twice the pixels of the others per repetition)
extern "C" int p3_chain_probe(const void* kv, const void* q, void* sink, int reps, float sl2e, int variant, int bf16, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 0: return bf16 ? launch_probe<__bf16, 1, 8, 0>(kv, q, sink, reps, sl2e, st) : launch_probe<_Float16, 1, 8, 0>(kv, q, sink, reps, sl2e, st);
    case 1: return bf16 ? launch_probe<__bf16, 1, 4, 0>(kv, q, sink, reps, sl2e, st) : launch_probe<_Float16, 1, 4, 0>(kv, q, sink, reps, sl2e, st);
    case 2: return bf16 ? launch_probe<__bf16, 2, 4, 1>(kv, q, sink, reps, sl2e, st) : launch_probe<_Float16, 2, 4, 1>(kv, q, sink, reps, sl2e, st);
    case 3: return bf16 ? launch_probe<__bf16, 1, 8, 1>(kv, q, sink, reps, sl2e, st) : launch_probe<_Float16, 1, 8, 1>(kv, q, sink, reps, sl2e, st);
    case 5: return bf16 ? launch_probe<__bf16, 1, 8, 2>(kv, q, sink, reps, sl2e, st) : launch_probe<_Float16, 1, 8, 2>(kv, q, sink, reps, sl2e, st);
    case 6: return bf16 ? launch_probe<__bf16, 1, 8, 3>(kv, q, sink, reps, sl2e, st) : launch_probe<_Float16, 1, 8, 3>(kv, q, sink, reps, sl2e, st);
    case 7: return bf16 ? launch_probe<__bf16, 1, 4, 2>(kv, q, sink, reps, sl2e, st) : launch_probe<_Float16, 1, 4, 2>(kv, q, sink, reps, sl2e, st);
    case 8: return bf16 ? launch_probe<__bf16, 1, 8, 4>(kv, q, sink, reps, sl2e, st) : launch_probe<_Float16, 1, 8, 4>(kv, q, sink, reps, sl2e, st);
    case 9: return bf16 ? launch_probe<__bf16, 1, 8, 5>(kv, q, sink, reps, sl2e, st) : launch_probe<_Float16, 1, 8, 5>(kv, q, sink, reps, sl2e, st);
    case 4: return bf16 ? launch_probe<__bf16, 2, 8, 1>(kv, q, sink, reps, sl2e, st) : launch_probe<_Float16, 2, 8, 1>(kv, q, sink, reps, sl2e, st);
  }
  return -1;
}
