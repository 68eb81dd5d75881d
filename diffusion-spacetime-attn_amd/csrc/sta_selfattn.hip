// sta_selfattn.hip — flash-style self-attention (attn1 of BasicTransformerBlock) for gfx950.
//
// Reference: CrossAttention.forward with context = x (ldm/modules/attention.py:175-197, called at :274):
//   sim = einsum(q, k) * scale; attn = softmax(sim); out = einsum(attn, v)   over N = H*W keys,
// which materialises [16, N, N] scores (N = 4096 at 512^2: 537 MB in fp16). SURVEY.md §8f ranks this the
// next component after the fused cross-attention: it shares the same transformer block and the same MFMA
// tile machinery. The differentiable path adds the log-sum-exp output (p.lse) for sta_selfattn_bwd.hip.
//
// Same "pixel is the MFMA column" layout as sta_xattn.hip (16x16x32 MFMA, lane = 16g + c):
//   S^T[key][px]  = K[key][:] . Q[px][:]       A = K rows (16 B per lane straight from the [B][N][ld] rows)
//   O^T[dcol][px] = V^T[dcol][:] . P^T[:][px]   A = V^T rows (16 B per lane from a TRANSPOSED V, [B][C][N])
// V^T costs nothing extra: the host computes it as W_v . x^T instead of x . W_v^T (one GEMM either way).
// The rows of an S^T tile are assigned to keys so that a lane ends up holding 8 CONSECUTIVE keys of a
// 32-key step (tile T, row 4g+r  <->  key 32(T>>1) + 8g + 4(T&1) + r): the softmax output is then directly
// the B operand of the PV product and the V^T fragment is one contiguous 16-byte load per lane.
// Online softmax over 64-key blocks; K/V^T fragments of a block are copied to LDS once per workgroup by
// LDS-DMA with per-lane source addresses (global_load_lds_dwordx4) and double-buffered.

#include "sta_selfattn_dev.h"

namespace {

struct SParams {
  const void* q;    // [B][N][ldq]
  const void* k;    // [B][N][ldk]
  const void* vt;   // V transposed: element (b, c, n) at b*vt_bs + c*vt_rs + n
  void* out;        // [B][N][C]
  int B, N, C, H, d, ldq, ldk;
  long vt_rs, vt_bs;   // row (channel) and batch strides of vt in elements: [B][C][N] -> (N, C*N); [C][B*N] -> (B*N, N)
  float sl2e;       // scale * log2(e)
  float* lse;       // optional [B][H][N]: log2-domain log-sum-exp of the scaled scores (what the backward kernels re-derive P from)
  int sfrag;        // 1: `out` leaves in self-attention out-fragment order (sta_selfattn_fwd_sfrag; d = 40, 8 heads, N % 16 == 0)
  unsigned* flags;  // sta_selfattn_fwd_optimistic: two state words, then one word per workgroup of the pipelined kernel — written by the
                    // optimistic launch (1: a denominator left its range), read by the repair launch of the standard loop (0: return at once)
  int optimistic;   // 1: launch the optimistic loop + the repair launch (the pipelined kernel's shapes, both 16-bit types)
};

extern __shared__ __attribute__((aligned(16))) char smem_sa[];

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

// SUMROW (head dims that are not a multiple of 16, e.g. 40): the first PADDED row of V^T is set to ones in
// registers, so the PV MFMAs accumulate the softmax denominator as row d of O^T — 16 v_add per 64 keys per query
// tile less in a kernel whose VALU pipe is 73 % busy; the running rescale covers it like any other row.
// PRE (q arrives pre-multiplied by scale*log2(e), p.sl2e == 1 — the host folds the factor into W_q): the running maximum
// is the INITIAL VALUE of the S^T accumulators, so the MFMAs deliver s - m and the exponent needs no multiply-subtract:
// 2.5 instead of 3.5 vector instructions per score in a loop whose time is the vector pipe's.
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }   // one v_max3_f32

// Normalise by the softmax denominator and store: row-major [B][N][C], or self-attention out-fragment order (p.sfrag); the log-sum-exp
// for the backward when asked. Shared by the loop below and the software-pipelined loop of the level-0 shape.
template <typename T, int NDT, int QT, bool SUMROW>
__device__ __forceinline__ void sa_epilogue(const SParams& p, f32x4 (&o)[QT][NDT], const float (&mrun)[QT], const float (&lrun)[QT],
                                            const int b, const int h, const int px0, const int g, const int c16, const int lane) {
  using V4 = typename Tr<T>::V4;
  const int N = p.N, C = p.C, d = p.d;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int px = px0 + 16 * qt + c16;
    float l;
    if constexpr (SUMROW) {   // denominator = row d of O^T: tile NDT-1, row d % 16 = 4*g_s + r_s, held by the lanes of lane row g_s
      const int rs_ = d & 15, g_s = rs_ >> 2, r_s = rs_ & 3;
      float lr = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) lr = (r == r_s) ? o[qt][NDT - 1][r] : lr;
      l = __shfl(lr, 16 * g_s + c16);
    } else {
      l = bfly_sum(lrun[qt]);
    }
    const float inv = 1.0f / l;
    if constexpr (NDT == 3) {
      if (p.sfrag) {
        // Out-fragment order for sta_to_out_ln_ofrag (csrc/sta_rowgemm.hip): the wave's 16 pixels are one row group; head h's
        // dims 0..31 are fragment h (lane (g, c): O^T rows 4g.. of tile 0 | tile 1 of pixel c — the accumulators as they are),
        // its dims 32..39 one lane row of fragment 8 + h / 4 (lane rows 0, 1 store 8 bytes each). d = 40, N % 16 == 0.
        if (px0 + 16 * qt >= N) continue;
        char* gb = (char*)p.out + ((size_t)b * N + px0 + 16 * qt) * C * sizeof(T);
        typename Tr<T>::V8 x8;
#pragma unroll
        for (int r = 0; r < 4; ++r) { x8[r] = (T)(o[qt][0][r] * inv); x8[4 + r] = (T)(o[qt][1][r] * inv); }
        *(typename Tr<T>::V8*)(gb + h * FRAG + lane * 16) = x8;
        if (g < 2) {
          V4 t4;
#pragma unroll
          for (int r = 0; r < 4; ++r) t4[r] = (T)(o[qt][2][r] * inv);
          *(V4*)(gb + (8 + (h >> 2)) * FRAG + (16 * (h & 3) + c16) * 16 + 8 * g) = t4;
        }
        if (p.lse && g == 0 && px < N) p.lse[((size_t)b * p.H + h) * N + px] = mrun[qt] * p.sl2e + __builtin_amdgcn_logf(l);
        continue;
      }
    }
    if (px >= N) continue;
    if (p.lse && g == 0)                   // P = exp2(s * sl2e - lse): exact whether or not the running maximum is stale
      p.lse[((size_t)b * p.H + h) * N + px] = mrun[qt] * p.sl2e + __builtin_amdgcn_logf(l);
    T* ob = (T*)p.out + ((size_t)b * N + px) * C + h * d;
#pragma unroll
    for (int u = 0; u < NDT; ++u) {
      const int dd = 16 * u + 4 * g;
      if (dd < d) {
        V4 r4;
#pragma unroll
        for (int r = 0; r < 4; ++r) r4[r] = (T)(o[qt][u][r] * inv);
        *(V4*)(ob + dd) = r4;
      }
    }
  }
}

// NW waves per workgroup. NW = 4 (QT query tiles of 16 per wave): the shipped geometry. NW = 8 with QT = 1: the same loop sized for FOUR
// waves per SIMD (<= 128 registers; two 512-thread workgroups per CU share each K / V^T block among 128 queries as before),
// built to test whether a fourth wave fills the third of the vector-issue port that three waves leave idle
// (profiles/r03_selfattn_32x32.md): it does not (launch_sa_cfg below).
template <typename T, int NKS, int NDT, int QT, bool SUMROW, bool PRE, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? (QT == 1 ? 4 : 2) : 1) void selfattn_fwd_kernel(const SParams p) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  constexpr int NKF = 4 * NKS;            // K fragments per block: 4 key tiles x NKS head-dim steps
  constexpr int NVF = 2 * NDT;            // V^T fragments per block: 2 key steps x NDT head-dim tiles
  constexpr int NFR = NKF + NVF;
  constexpr int PER = (NFR + NW - 1) / NW; // LDS-DMA instructions per wave per block (same for every wave: the
  constexpr int NFRP = NW * PER;          //   counted vmcnt below needs it) -> NFRP - NFR padding copies
  constexpr int BB = NFRP * FRAG;         // LDS bytes per block buffer (padding copies land in its tail)
  constexpr int DEPTH = 3;                // blocks resident in LDS: compute blk while blk+1, blk+2 are in flight
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int N = p.N, C = p.C, d = p.d;
  const int b = blockIdx.y;
  int tile, h;
  if (p.H == 8) { tile = blockIdx.x >> 3; h = blockIdx.x & 7; } else { tile = blockIdx.x / p.H; h = blockIdx.x % p.H; }
  const int px0 = (tile * NW + wv) * 16 * QT;

  const T* qb = (const T*)p.q + (size_t)b * N * p.ldq + h * d;
  const T* kb = (const T*)p.k + (size_t)b * N * p.ldk + h * d;
  const T* vb = (const T*)p.vt + (size_t)b * p.vt_bs + (size_t)(h * d) * p.vt_rs;

  // Q (B operand), zero in the padded head-dim slots so that K's padding never matters
  V8 qf[QT][NKS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int px = px0 + 16 * qt + c16;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      V8 z = {};
      const int dd = 32 * s + 8 * g;
      qf[qt][s] = (px < N && dd < d) ? *(const V8*)(qb + (size_t)px * p.ldq + dd) : z;
    }
  }

  // LDS-DMA of one 64-key block: every wave copies fragments wv, wv+4, ... ; per-lane source addresses,
  // lane-linear destination. Out-of-range rows/keys are clamped to valid memory (their products are
  // multiplied by zero Q slots, masked scores or discarded output rows). The per-lane source pointers are
  // computed ONCE and advanced by a constant per block (64 key rows of K, 64 keys along a V^T row): the
  // address arithmetic of a naive per-block recomputation was most of the loop's instruction count.
  const char* src[PER];
  int step[PER];                           // bytes per block
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int f = wv + NW * i;
    const int fs = f < NFR ? f : 0;       // padding copy: re-read fragment 0 into the unused tail slot
    if (fs < NKF) {
      const int Tt = fs / NKS, s = fs - Tt * NKS;
      src[i] = (const char*)(kb + (size_t)min(tile_key(Tt, c16), N - 1) * p.ldk + min(32 * s + 8 * g, d - 8));
      step[i] = __builtin_amdgcn_readfirstlane(KB * p.ldk * (int)sizeof(T));     // wave-uniform -> SGPR
    } else {
      const int f2 = fs - NKF;
      const int s2 = f2 / NDT, u = f2 - s2 * NDT;
      src[i] = (const char*)(vb + (size_t)min(16 * u + c16, d - 1) * p.vt_rs + min(32 * s2 + 8 * g, N - 8));
      step[i] = __builtin_amdgcn_readfirstlane(KB * (int)sizeof(T));
    }
  }
  // SUMROW: the lanes that feed row d of the last head-dim tile (the ones row whose PV product is the softmax denominator)
  // are switched off in the DMA of their V^T fragments; their 16 bytes of every ring slot are written with ones ONCE here.
  // (Selecting the ones in registers cost 8 v_cndmask per block: 2018-2022 vs 2025-2039 us, B = 64, N = 4096, d = 40.)
  unsigned keep = ~0u;                     // bit i: this lane takes part in the wave's i-th DMA
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if constexpr (SUMROW) {
      const int f = wv + NW * i, f2 = f - NKF;
      const bool ones_frag = f < NFR && f2 >= 0 && f2 % NDT == NDT - 1;
      if (ones_frag && 16 * (NDT - 1) + c16 == d) {
        keep &= ~(1u << i);
        V8 ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;
#pragma unroll
        for (int sl = 0; sl < DEPTH; ++sl) *(V8*)(smem_sa + sl * BB + f * FRAG + lane * 16) = ones;
      }
    }
  }
  const int nfull = N / KB;                // blocks whose 64 keys all exist (the incremental pointers are exact)
  const int dm8 = d - 8, nm8 = N - 8;
  auto stage = [&, N, d, dm8, nm8](int blk, char* dst) __attribute__((always_inline)) {   // N, d by value: a by-reference capture ended up in scratch
    if (blk < nfull) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        if (keep >> i & 1)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                           (__attribute__((address_space(3))) void*)(dst + (wv + NW * i) * FRAG), 16, 0, 0);
        src[i] += step[i];
      }
      return;
    }
    const int k0 = blk * KB;               // partial last block (N % 64 != 0): clamp every key
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wv + NW * i;
      const int fs = f < NFR ? f : 0;
      const T* sp;
      if (fs < NKF) {
        const int Tt = fs / NKS, s = fs - Tt * NKS;
        sp = kb + (size_t)min(k0 + tile_key(Tt, c16), N - 1) * p.ldk + min(32 * s + 8 * g, dm8);
      } else {
        const int f2 = fs - NKF;
        const int s2 = f2 / NDT, u = f2 - s2 * NDT;
        sp = vb + (size_t)min(16 * u + c16, d - 1) * p.vt_rs + min(k0 + 32 * s2 + 8 * g, nm8);
      }
      if (keep >> i & 1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                         (__attribute__((address_space(3))) void*)(dst + f * FRAG), 16, 0, 0);
    }
  };

  f32x4 o[QT][NDT];
  float mrun[QT], lrun[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrun[qt] = PRE ? 0.f : -3.0e38f;      // PRE: block 0 always takes the exact maximum (see below)
    lrun[qt] = 0.f;
#pragma unroll
    for (int u = 0; u < NDT; ++u) o[qt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // 3-deep LDS ring, one barrier per block. An LDS-DMA is a pending write on the VM counter, so the wait is a
  // COUNTED vmcnt (the PER newest DMAs — block blk+1 — may stay in flight) and the barrier is the raw s_barrier:
  // __syncthreads() would emit vmcnt(0) and drain the ring (cdna_hip_programming.md, "Pipelining across barriers").
  const int nblk = (N + KB - 1) / KB;
  stage(0, smem_sa);
  if (nblk > 1) stage(1, smem_sa + BB);
  // The ring slot of a block is a COMPILE-TIME constant: the loop body is instantiated once per slot and the loop walks three
  // blocks per trip. Every fragment read is then `ds_read_b128 v, lane16 offset:SLOT * BB + f * 1024` — with a run-time slot hipcc
  // spent two vector adds per fragment read on the address (30 of the ~130 vector instructions of a block in a loop that is
  // vector-issue bound: profiles/r02_pmc_selfattn.md). Same box, B = 64, N = 4096, d = 40: 2114-2124 -> 2059-2064 us.
  const char* lane_base = smem_sa + lane * 16;
  auto body = [&](auto slot_tag, const int blk) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    if (blk + 1 < nblk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // block `blk` landed for every wave; everyone left block blk-1 (the loop WITHOUT
                                                       // this rendezvous — wrong results, timing only — is 3 % slower: 2108 vs 2049 us)
    if (blk + 2 < nblk) stage(blk + 2, smem_sa + ((SLOT + 2) % DEPTH) * BB);
    const bool tail = (blk + 1) * KB > N;              // partial last block: mask the missing keys
    const V8* fr = (const V8*)(lane_base + SLOT * BB);
    V8 ka[NKF];
#pragma unroll
    for (int f = 0; f < NKF; ++f) ka[f] = fr[f * 64];
    f32x4 st[QT][4];
    __builtin_amdgcn_s_setprio(1);      // MFMA clusters at raised priority: +2 % (1405 -> 1377 us at B = 32, N = 4096, d = 40)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float i0 = PRE ? -mrun[qt] : 0.f;       // (a loop-carried {-m,-m,-m,-m} vector as C operand measured the same: 2190-2259 vs 2233-2276 us)
        st[qt][t] = Tr<T>::mfma(ka[t * NKS], qf[qt][0], f32x4{i0, i0, i0, i0});
#pragma unroll
        for (int s = 1; s < NKS; ++s) st[qt][t] = Tr<T>::mfma(ka[t * NKS + s], qf[qt][s], st[qt][t]);
      }
    __builtin_amdgcn_s_setprio(0);
    V8 va[NVF];
#pragma unroll
    for (int f = 0; f < NVF; ++f) va[f] = fr[(NKF + f) * 64];
    V8 pb[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      // (Measured and rejected: scores pre-scaled with packed multiplies + v_max3 + packed subtract/add — 25 % fewer
      // VALU instructions, 7 % SLOWER (784 vs 735 us at B=16, N=4096, d=40): packed fp32 VALU next to MFMAs is an
      // anti-lever on this chip, MI355X_MICROARCH.md "price of one filler beside MFMAs".)
      if (tail) {
        // a real branch, taken only for a partial last block: the empty volatile asm keeps hipcc from
        // if-converting it into 16 compare+select pairs that would execute on every block
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (blk * KB + tile_key(t, 4 * g + r) >= N) st[qt][t][r] = -3.0e38f;
      }
      // this LANE's maximum over its 16 scores: eight v_max3 / v_max. The reduction over the four lane rows that share a
      // pixel only runs when a rescale is due: "some lane saw more than the threshold" is the same wave-wide condition as
      // "some pixel's maximum did" and needs no cross-lane step (2 x (mov, permlane swap, max) per q tile and block):
      // 2036-2049 -> 2018-2022 us.
      float bm = fmaxf(max3f(max3f(st[qt][0][0], st[qt][0][1], st[qt][0][2]), max3f(st[qt][0][3], st[qt][1][0], st[qt][1][1]),
                             max3f(st[qt][1][2], st[qt][1][3], st[qt][2][0])),
                       max3f(max3f(st[qt][2][1], st[qt][2][2], st[qt][2][3]), max3f(st[qt][3][0], st[qt][3][1], st[qt][3][2]),
                             st[qt][3][3]));
      if constexpr (!PRE) bm = bfly_max(bm);
      // Deferred rescale: the running maximum (and with it the O^T accumulators, 12 multiplies + an exp per tile and
      // block) is only moved when some pixel of the wave saw its maximum grow by more than 2^RESCALE_LOG2 since the last
      // move; until then P = exp2(s - stale max) may reach 2^RESCALE_LOG2 = 256, well inside fp16 / bf16, and the
      // ones row of V^T (or lrun) sums the same P, so the ratio O / l is unchanged. Wave-uniform branch. Measured
      // 1385 -> 1344 us at B = 32, N = 4096, d = 40 (neutral at d = 80); both sides of the branch are forced in
      // tests/test_kernel_gpu.py::test_self_attention_deferred_rescale_branches.
      float rs = 0.f;
      if constexpr (PRE) {
        // bm = max(s) - m (the accumulators started at -m). Block 0 moves m to the exact maximum whatever its sign (m starts
        // at 0 and O^T, l at 0: nothing to rescale); later blocks only when some pixel's maximum grew by more than 2^RESCALE_LOG2.
        const bool first = blk == 0;
        if (first || __any(bm > RESCALE_LOG2)) {
          bm = bfly_max(bm);
          const float delta = first ? bm : fmaxf(bm, 0.f);
          mrun[qt] += delta;
          if (!first) {
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            if constexpr (!SUMROW) lrun[qt] *= alpha;
#pragma unroll
            for (int u = 0; u < NDT; ++u) o[qt][u] *= alpha;
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[qt][t][r] -= delta;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(st[qt][t][r]);
            st[qt][t][r] = e;
            if constexpr (!SUMROW) rs += e;
          }
      } else {
      if (__any((bm - mrun[qt]) * p.sl2e > RESCALE_LOG2)) {
        const float mnew = fmaxf(mrun[qt], bm);
        const float alpha = __builtin_amdgcn_exp2f((mrun[qt] - mnew) * p.sl2e);
        mrun[qt] = mnew;
        if constexpr (!SUMROW) lrun[qt] *= alpha;
#pragma unroll
        for (int u = 0; u < NDT; ++u) o[qt][u] *= alpha;
      }
      const float mnew = mrun[qt];
      const float off = mnew * p.sl2e;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qt][t][r], p.sl2e, -off));
          st[qt][t][r] = e;
          if constexpr (!SUMROW) rs += e;
        }
      }
      if constexpr (!SUMROW) lrun[qt] += rs;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[qt][s2][j] = (T)st[qt][2 * s2 + (j >> 2)][j & 3];
    }
    // O^T += V^T P^T
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < NDT; ++u)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][u] = Tr<T>::mfma(va[s2 * NDT + u], pb[qt][s2], o[qt][u]);
    __builtin_amdgcn_s_setprio(0);
  };
  for (int blk = 0; blk < nblk; blk += DEPTH) {
    body(std::integral_constant<int, 0>{}, blk);
    if (blk + 1 < nblk) body(std::integral_constant<int, 1>{}, blk + 1);
    if (blk + 2 < nblk) body(std::integral_constant<int, 2>{}, blk + 2);
  }

  sa_epilogue<T, NDT, QT, SUMROW>(p, o, mrun, lrun, b, h, px0, g, c16, lane);
}


// ---- the level-0 shape (d = 40: NKS = 2, NDT = 3; q in log2 units; ones row; N % 64 == 0), software-pipelined -----------------
//
// The loop above runs S^T MFMAs -> softmax -> PV MFMAs one after the other inside a wave and leaves the overlap of the matrix pipe and
// the vector pipe to chance (which phase the SIMD's other waves happen to be in): counters showed the two pipes' busy times simply ADD
// UP — a 64-key block takes 1066 cycles against 476 of matrix pipe and 488 of vector issue (profiles/r03_selfattn_32x32.md). Here
// one wave carries the overlap itself: its two query tiles A and B run half a block apart, and every half step issues the MFMAs of one
// tile with the softmax of the OTHER between them, one small group of vector instructions behind each MFMA (an MFMA of this shape
// leaves room for two plain or one transcendental instruction in its shadow, profiles/r03_mfma_valu_coissue.txt):
//
//     X(j):  MFMAs  O_A += V^T(j-1) P_A(j-1)   S_A(j) = K(j) Q_A - m_A      ||   vector  P_B(j-1) = exp2(S_B(j-1)), running maximum of B
//     Y(j):  MFMAs  O_B += V^T(j-1) P_B(j-1)   S_B(j) = K(j) Q_B - m_B      ||   vector  P_A(j)   = exp2(S_A(j)),   running maximum of A
//
// 14 MFMAs and one tile's softmax (8 max3, 16 exp2, 8 convert) per half. K(j) and V^T(j-1) are read from the LDS ring once per
// block for both tiles; the ring is 4 deep (block j-1 is still being read while j+1, j+2 are in flight): 64 KiB, two workgroups per CU.
// The K operands of one 64-key block for d = 40: per key tile one k = 32 operand (dims 0..31) and one k = 16 operand (dims 32..39; its
// upper eight k slots meet zeros on the q side). The k = 16 MFMA costs what the k = 32 one does, but the second k = 32 step of the
// plain loop carried 24 padded dims per key through L2 -> LDS and LDS -> registers: 11 instead of 14 KiB per block in a launch
// whose LDS-DMA stream is what it waits for (profiles/r05_selfattn.md).
template <typename T> struct SaK {
  typename Tr<T>::V8 big[4];
  typename Tr<T>::V4 sm[4];
};
template <typename T> struct SaM16;
template <> struct SaM16<_Float16> {
  static __device__ __forceinline__ f32x4 mfma(f16x4 a, f16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};
template <> struct SaM16<__bf16> {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  static __device__ __forceinline__ f32x4 mfma(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
};
template <typename T> struct SaQ {          // one query tile's B operands: dims 0..31, and dims 32..39 | zeros
  typename Tr<T>::V8 big;
  typename Tr<T>::V4 sm;
};

template <typename T, bool OPT>
__device__ __forceinline__ void sa_softmax_chunk(const int k, f32x4 (&s)[4], f32x4 (&o)[3], typename Tr<T>::V8 (&pb)[2], float& m, float (&tmp)[6],
                                                 const bool first) {
  // the vector work of one tile's softmax, cut into the pieces that go behind MFMA number k of a half step (k = 0 .. 13)
  // OPT (sta_selfattn_fwd_optimistic): no running maximum behind a tile's first block — P = exp2(S - m_0) with the first block's exact
  // maximum; the kernel checks the range of every denominator at the end and flags the workgroup for the repair launch
  if (OPT && k < 4 && !first) return;
  switch (k) {
    case 0: tmp[0] = max3f(s[0][0], s[0][1], s[0][2]); tmp[1] = max3f(s[0][3], s[1][0], s[1][1]); break;
    case 1: tmp[2] = max3f(s[1][2], s[1][3], s[2][0]); tmp[3] = max3f(s[2][1], s[2][2], s[2][3]); break;
    case 2: tmp[4] = max3f(s[3][0], s[3][1], s[3][2]); tmp[0] = max3f(tmp[0], tmp[1], tmp[2]); tmp[3] = max3f(tmp[3], tmp[4], s[3][3]); break;
    case 3: {
      float bm = fmaxf(tmp[0], tmp[3]);          // this lane's maximum of s - m over its 16 scores
      // deferred rescale (see the loop above): the running maximum moves only when some pixel of the wave saw more than 2^RESCALE_LOG2
      if (first || __any(bm > RESCALE_LOG2)) {
        bm = bfly_max(bm);
        const float delta = first ? bm : fmaxf(bm, 0.f);
        m += delta;
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int u = 0; u < 3; ++u) o[u] *= alpha;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[t][r] -= delta;
      }
      break;
    }
    default: {
      if (k >= 4 && k < 12) {                    // two exponentials behind each of MFMAs 4 .. 11
        const int t = (k - 4) >> 1, r0 = 2 * ((k - 4) & 1);
        s[t][r0] = __builtin_amdgcn_exp2f(s[t][r0]);
        s[t][r0 + 1] = __builtin_amdgcn_exp2f(s[t][r0 + 1]);
      }
      if (k >= 6 && (k & 1) == 0) {              // tile t's four values are final two MFMAs later: convert them (k = 6, 8, 10, 12)
        const int t = (k - 6) >> 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) pb[t >> 1][4 * (t & 1) + r] = (T)s[t][r];
      }
    }
  }
}

#ifndef STA_SA_ABLATE
#define STA_SA_ABLATE 0     // timing experiments (tools/asm_patch_ab.py flag:-DSTA_SA_ABLATE=bits; wrong results): 1 no softmax, 2 no MFMAs,
#endif                      // 4 no per-block rendezvous, 8 no per-block LDS reads, 16 no LDS-DMA inside the loop
// one half step: 14 MFMAs of tile `f` (PV of block j-1 when DO_PV, S^T of block j when DO_S) with the softmax of tile `v` between them
template <typename T, bool DO_PV, bool DO_S, bool DO_SM, bool OPT = false>
__device__ __forceinline__ void sa_half_step(const SaK<T>& ka, const typename Tr<T>::V8 (&va)[6], const SaQ<T>& qf,
                                             f32x4 (&s_f)[4], f32x4 (&o_f)[3], const typename Tr<T>::V8 (&p_f)[2], const float m_f,
                                             f32x4 (&s_v)[4], f32x4 (&o_v)[3], typename Tr<T>::V8 (&p_v)[2], float& m_v, const bool first_v) {
  float tmp[6];
  int k = 0;
  auto vec = [&]() __attribute__((always_inline)) {
#if !(STA_SA_ABLATE & 1)
    if constexpr (DO_SM) sa_softmax_chunk<T, OPT>(k, s_v, o_v, p_v, m_v, tmp, first_v);
#else
    if (DO_SM && k == 0) {          // keep the scores and the P operands alive without the vector work
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(s_v[t]));
      asm volatile("" : "+v"(p_v[0]), "+v"(p_v[1]));
    }
#endif
    ++k;
    __builtin_amdgcn_sched_barrier(0);
  };
  if constexpr (DO_PV) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int u = 0; u < 3; ++u) {
#if !(STA_SA_ABLATE & 2)
        o_f[u] = Tr<T>::mfma(va[s2 * 3 + u], p_f[s2], o_f[u]);
#else
        asm volatile("" : "+v"(o_f[u]) : "v"(va[s2 * 3 + u]), "v"(p_f[s2]));
#endif
        vec();
      }
  }
  if constexpr (DO_S) {
    const float i0 = -m_f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#if !(STA_SA_ABLATE & 2)
      s_f[t] = Tr<T>::mfma(ka.big[t], qf.big, f32x4{i0, i0, i0, i0});
#else
      s_f[t] = f32x4{i0, i0, i0, i0};
      asm volatile("" : "+v"(s_f[t]) : "v"(ka.big[t]), "v"(qf.big));
#endif
      vec();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#if !(STA_SA_ABLATE & 2)
      s_f[t] = SaM16<T>::mfma(ka.sm[t], qf.sm, s_f[t]);       // three other MFMAs behind its k = 32 step: no mixed-shape hazard (sta/isa_lint.py)
#else
      asm volatile("" : "+v"(s_f[t]) : "v"(ka.sm[t]), "v"(qf.sm));
#endif
      vec();
    }
  }
  if constexpr (DO_SM) {
    while (k < 14) vec();                        // a half step without one of the MFMA groups (first / last block): the rest of the softmax
  }
}

constexpr unsigned SA_OPT_SITOUT = 64;
constexpr unsigned SA_FLAG0 = 32;      // the per-workgroup flag words start one 128-byte line behind the two state words (which only atomics touch)
template <typename T, int NW, int QT, bool OPT = false>
__global__ __launch_bounds__(64 * NW, 2) void selfattn_fwd_pipe_kernel(const SParams p) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  // flags (sta_selfattn_fwd_optimistic): word 0 = calls the optimistic loop still sits out, word 1 = workgroups flagged by this call,
  // word 32 + w = workgroup w of this grid must be redone (the state words have a 128-byte line to themselves). The two leading words carry state from call to call: a call in which more than
  // an eighth of the workgroups failed (activations whose row maxima lie further than the type's headroom above the own neighbourhood's)
  // switches the optimistic loop off for the next SA_OPT_SITOUT calls — those cost the standard loop plus an empty launch — instead of
  // paying both loops every time.
  const unsigned wg_id = blockIdx.y * gridDim.x + blockIdx.x;
  if constexpr (!OPT) {      // the repair launch behind an optimistic one: only flagged workgroups run
    if (p.flags) {
      const unsigned mine = p.flags[SA_FLAG0 + wg_id];
      if (wg_id == 0 && threadIdx.x == 0) {           // the call's bookkeeping (the optimistic launch of the next call is stream-ordered behind this one)
        // (device-scope atomics: the optimistic launch read word 0 through the scalar cache, and a plain load here was served the line it
        // left there — word 1 without the failures counted since)
        const unsigned failed = atomicExch(p.flags + 1, 0u);
        const unsigned sitout = __hip_atomic_load(p.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.flags, 8u * failed > gridDim.x * gridDim.y ? SA_OPT_SITOUT : (sitout ? sitout - 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (mine == 0) return;
    }
  } else {
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) {      // sitting out: everything goes to the standard loop
      if (threadIdx.x == 0) p.flags[SA_FLAG0 + wg_id] = 1u;
      return;
    }
  }
  // fragments of a block (1 KiB each): 0..3 K dims 0..31 of key tile t; 4 K dims 32..39 of all 64 keys (lane = key); 5..10 V^T
  constexpr int NDT = 3, NKF = 5, NVF = 6, NFR = NKF + NVF, PER = (NFR + NW - 1) / NW, NFRP = NW * PER, BB = NFRP * FRAG, DEPTH = 4;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int N = p.N, d = p.d;
  const int b = blockIdx.y;
  const int tile = blockIdx.x >> 3, h = blockIdx.x & 7;
  const int px0 = (tile * NW + wv) * 16 * QT;
  const T* qb = (const T*)p.q + (size_t)b * N * p.ldq + h * d;
  const T* kb = (const T*)p.k + (size_t)b * N * p.ldk + h * d;
  const T* vb = (const T*)p.vt + (size_t)b * p.vt_bs + (size_t)(h * d) * p.vt_rs;
  SaQ<T> qf[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int px = px0 + 16 * qt + c16;
    V8 z = {};
    V4 z4 = {};
    qf[qt].big = px < N ? *(const V8*)(qb + (size_t)px * p.ldq + 8 * g) : z;
    qf[qt].sm = (px < N && g < 2) ? *(const V4*)(qb + (size_t)px * p.ldq + 32 + 4 * g) : z4;     // k slots 8..15 of the k = 16 step: zeros
  }
  // LDS-DMA of a 64-key block as in the loop above (fragments wv, wv + NW, ...; per-lane sources advanced by a constant)
  const char* src[PER];
  int step[PER];
  unsigned keep = ~0u;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int f = wv + NW * i;
    const int fs = f < NFR ? f : 0;
    if (fs < 4) {
      src[i] = (const char*)(kb + (size_t)tile_key(fs, c16) * p.ldk + 8 * g);
      step[i] = __builtin_amdgcn_readfirstlane(KB * p.ldk * (int)sizeof(T));
    } else if (fs == 4) {
      src[i] = (const char*)(kb + (size_t)lane * p.ldk + 32);           // dims 32..39 of key `lane`: 16 bytes at LDS offset 16 lane
      step[i] = __builtin_amdgcn_readfirstlane(KB * p.ldk * (int)sizeof(T));
    } else {
      const int f2 = fs - NKF;
      const int s2 = f2 / NDT, u = f2 - s2 * NDT;
      src[i] = (const char*)(vb + (size_t)min(16 * u + c16, d - 1) * p.vt_rs + 32 * s2 + 8 * g);
      step[i] = __builtin_amdgcn_readfirstlane(KB * (int)sizeof(T));
    }
    const int f2 = f - NKF;
    if (f < NFR && f2 >= 0 && f2 % NDT == NDT - 1 && 16 * (NDT - 1) + c16 >= d) {
      // rows 40..47 of V^T do not exist: row 40 is the ones row (softmax denominator), rows 41..47 feed accumulator rows nobody reads.
      // Their lanes take no part in the DMA (7/16 of two fragments less through L2 -> LDS); their LDS bytes are written once per ring slot
      keep &= ~(1u << i);
      V8 fill;
#pragma unroll
      for (int j = 0; j < 8; ++j) fill[j] = (T)(16 * (NDT - 1) + c16 == d ? 1.0f : 0.0f);
#pragma unroll
      for (int sl = 0; sl < DEPTH; ++sl) *(V8*)(smem_sa + sl * BB + f * FRAG + lane * 16) = fill;
    }
  }
  // 11 fragments over NW waves: the first NFR - NW (PER - 1) waves copy PER of them per block, the others PER - 1 — no padding copies
  // (the loop above copies 16 KiB per block: 14 fragments + fragment 0 twice more to keep one vmcnt for every wave); each wave counts
  // its own copies (wave-uniform branch around the s_waitcnt immediates)
  const bool full_share = wv < NFR - NW * (PER - 1);
  const int nblk = N / KB;
  // OPT: the key loop starts at the workgroup's OWN 64-key block and wraps around (block number j of the loop = key block (j0 + j) mod
  // nblk). A query's largest logits sit in its own neighbourhood of the image, so the first block's exact maximum m_0 is within fp16's
  // 2^16 of the row maximum for ordinary activations and P = exp2(S - m_0) needs no running maximum in EITHER 16-bit type; whatever
  // does overflow is caught by the denominator test below and redone by the repair launch.
  int to_wrap = nblk;
  if constexpr (OPT) {
    const int j0 = min(tile * (16 * QT * NW / KB), nblk - 1);
    to_wrap = nblk - j0;
#pragma unroll
    for (int i = 0; i < PER; ++i) src[i] += (long)j0 * step[i];
  }
  auto stage = [&](char* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (i == PER - 1 && !full_share) break;
      if (keep >> i & 1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                         (__attribute__((address_space(3))) void*)(dst + (wv + NW * i) * FRAG), 16, 0, 0);
      src[i] += step[i];
    }
    if constexpr (OPT) {
      if (--to_wrap == 0) {
#pragma unroll
        for (int i = 0; i < PER; ++i) src[i] -= (long)nblk * step[i];
      }
    }
  };
  auto arrive = [&](const int blk) __attribute__((always_inline)) {      // block `blk` landed for every wave; everyone left block blk - 1
    if (blk + 1 < nblk) {
      if (full_share) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER - 1) : "memory");
    }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (no lgkmcnt wait: what this wave read from the slot that is overwritten next has been consumed by MFMAs a half step ago)
#if !(STA_SA_ABLATE & 4)
    __builtin_amdgcn_s_barrier();
#endif
  };
  // QT query tiles per wave, a 1 / QT block apart: step i of a block issues the MFMAs of tile i with the softmax of tile i - 1 (mod QT)
  // between them. QT = 3: 192 queries per workgroup share a block's 11 KiB (and every operand read from LDS serves three tiles).
  f32x4 o[QT][3], sc[QT][4];
  V8 pb[QT][2];
  float mrun[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrun[qt] = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) o[qt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const char* lane_base = smem_sa + lane * 16;
  // Operand registers are double-buffered: K(j+1) and V^T(j) are requested from LDS in the MIDDLE of block j (behind the rendezvous
  // for block j+1) and land under its second half step. With the reads at the top of a block every wave of the workgroup — they leave
  // the barrier together — sat out the LDS latency at the same time (profiles/r05_selfattn.md).
  SaK<T> ka[2];
  V8 va[2][NVF];
  // the k = 16 operand of key tile t: row c = key tile_key(t, c), k slots 4g..4g+3 = dims 32 + 4 (g & 1) .. (lane rows 2, 3 meet zeros on
  // the q side: they re-read rows 0, 1's bytes)
  int ksm_off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) ksm_off[t] = 4 * FRAG + 16 * tile_key(t, c16) + 8 * (g & 1);
  auto load_k = [&](SaK<T>& dst, const int slot) __attribute__((always_inline)) {
    const V8* fr = (const V8*)(lane_base + slot * BB);
#pragma unroll
    for (int t = 0; t < 4; ++t) dst.big[t] = fr[t * 64];
#pragma unroll
    for (int t = 0; t < 4; ++t) dst.sm[t] = *(const V4*)(smem_sa + slot * BB + ksm_off[t]);
  };
  auto load_v = [&](V8 (&dst)[NVF], const int slot) __attribute__((always_inline)) {
    const V8* fr = (const V8*)(lane_base + slot * BB);
#pragma unroll
    for (int f = 0; f < NVF; ++f) dst[f] = fr[(NKF + f) * 64];
  };
  stage(smem_sa);
  if (nblk > 1) stage(smem_sa + BB);
  // block 0: S of tile 0; rendezvous for block 1, its K and block 0's V^T requested; then S of tile i with the softmax of tile i - 1
  // between its MFMAs (every tile takes the exact maximum of its first block: `first`)
  arrive(0);
  if (nblk > 2) stage(smem_sa + 2 * BB);
  load_k(ka[0], 0);
  sa_half_step<T, false, true, false, OPT>(ka[0], va[0], qf[0], sc[0], o[0], pb[0], mrun[0], sc[QT - 1], o[QT - 1], pb[QT - 1], mrun[QT - 1], false);
  if (nblk > 1) {
    arrive(1);
    if (nblk > 3) stage(smem_sa + 3 * BB);
    load_k(ka[1], 1);
  }
  load_v(va[1], 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 1; i < QT; ++i)
    sa_half_step<T, false, true, true, OPT>(ka[0], va[0], qf[i], sc[i], o[i], pb[i], mrun[i], sc[i - 1], o[i - 1], pb[i - 1], mrun[i - 1], true);
  auto body = [&](auto slot_tag, const int blk) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value, c = SLOT & 1;      // block blk sits in ring slot SLOT = blk % 4, register set blk & 1
    sa_half_step<T, true, true, true, OPT>(ka[c], va[c], qf[0], sc[0], o[0], pb[0], mrun[0], sc[QT - 1], o[QT - 1], pb[QT - 1], mrun[QT - 1], blk == 1);
    if (blk + 1 < nblk) {
      arrive(blk + 1);
#if !(STA_SA_ABLATE & 16)
      if (blk + 3 < nblk) stage(smem_sa + ((SLOT + 3) % DEPTH) * BB);
#endif
      load_k(ka[c ^ 1], (SLOT + 1) % DEPTH);
    }
    load_v(va[c ^ 1], SLOT);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 1; i < QT; ++i)
      sa_half_step<T, true, true, true, OPT>(ka[c], va[c], qf[i], sc[i], o[i], pb[i], mrun[i], sc[i - 1], o[i - 1], pb[i - 1], mrun[i - 1], false);
  };
  for (int blk = 1; blk < nblk; blk += DEPTH) {
    body(std::integral_constant<int, 1>{}, blk);
    if (blk + 1 < nblk) body(std::integral_constant<int, 2>{}, blk + 1);
    if (blk + 2 < nblk) body(std::integral_constant<int, 3>{}, blk + 2);
    if (blk + 3 < nblk) body(std::integral_constant<int, 0>{}, blk + 3);
  }
  // drain: PV of the last block for tile 0 with the last tile's last softmax between, then PV for the others (V^T of the last block:
  // register set nblk & 1)
  auto drain = [&](auto set_tag) __attribute__((always_inline)) {
    constexpr int c = decltype(set_tag)::value;
    sa_half_step<T, true, false, true, OPT>(ka[c], va[c], qf[0], sc[0], o[0], pb[0], mrun[0], sc[QT - 1], o[QT - 1], pb[QT - 1], mrun[QT - 1], nblk == 1);
#pragma unroll
    for (int i = 1; i < QT; ++i)
      sa_half_step<T, true, false, false, OPT>(ka[c], va[c], qf[i], sc[i], o[i], pb[i], mrun[i], sc[i - 1], o[i - 1], pb[i - 1], mrun[i - 1], false);
  };
  if (nblk & 1) drain(std::integral_constant<int, 1>{});
  else drain(std::integral_constant<int, 0>{});
  float lrun[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) lrun[qt] = 0.f;
  if constexpr (OPT) {
    // every query's denominator (row 40 of O^T = tile 2, row 8: lane row 2, register 0) must lie in [2^-100, 2^100) for bf16 and in
    // [2^-100, 2^15) for fp16 — a P that reached fp16's largest number (as inf or saturated, whatever the conversion's rounding mode
    // does) alone lifts its denominator past 2^15; the first block's own maximum contributes exactly 1. Tested on the bits
    // (-ffinite-math-only would fold a class test of inf / NaN away); one word per workgroup tells the repair launch what to redo
    constexpr unsigned LO = 0x0D800000u, HI = std::is_same<T, __bf16>::value ? 0x71800000u : 0x47000000u;
    bool bad = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) bad |= (g == 2) && (__float_as_uint(o[qt][2][0]) - LO) >= (HI - LO);
    const int wg_bad = __syncthreads_or((int)bad);
    if (threadIdx.x == 0) {
      p.flags[SA_FLAG0 + wg_id] = wg_bad ? 1u : 0u;
      if (wg_bad) atomicAdd(p.flags + 1, 1u);
    }
  }
  sa_epilogue<T, 3, QT, true>(p, o, mrun, lrun, b, h, px0, g, c16, lane);
}

template <typename T, int NW, int QT>
int launch_sa_pipe(const SParams& p, hipStream_t st) {
  constexpr int lds = 4 * ((11 + NW - 1) / NW * NW) * FRAG;
  static StaLdsAttr attr, attr_opt;
  if (!attr.ensure((const void*)selfattn_fwd_pipe_kernel<T, NW, QT>, lds)) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(selfattn pipe) failed");
  const int tiles = (p.N + 16 * QT * NW - 1) / (16 * QT * NW);
  {
    if (p.optimistic) {
      // the optimistic loop (no running maximum behind a tile's first block: a tenth of the loop's instructions less), then the
      // standard loop for the workgroups it flagged — with ordinary logits every workgroup of the second launch returns at once
      if (!attr_opt.ensure((const void*)selfattn_fwd_pipe_kernel<T, NW, QT, true>, lds)) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(selfattn pipe) failed");
      hipLaunchKernelGGL((selfattn_fwd_pipe_kernel<T, NW, QT, true>), dim3(tiles * p.H, p.B), dim3(64 * NW), lds, st, p);
    }
  }
  hipLaunchKernelGGL((selfattn_fwd_pipe_kernel<T, NW, QT>), dim3(tiles * p.H, p.B), dim3(64 * NW), lds, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "selfattn pipe launch: %s", hipGetErrorString(e));
}

#ifdef STA_EXPERIMENT_SELFATTN32
#include "../../tools/experiments/selfattn_fwd32.inc"     // 32x32x16-MFMA variant: measured slower in wall time (lower clocks), tools-only
#endif

template <typename T, int NKS, int NDT, int QT, bool SUMROW, bool PRE, int NW>
int launch_sa_geom(const SParams& p, hipStream_t st) {
  constexpr int lds = 3 * (((4 * NKS + 2 * NDT) + NW - 1) / NW * NW) * FRAG;
  static StaLdsAttr attr;
  if (!attr.ensure((const void*)selfattn_fwd_kernel<T, NKS, NDT, QT, SUMROW, PRE, NW>, lds)) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(selfattn) failed");
  const int tiles = (p.N + 16 * NW * QT - 1) / (16 * NW * QT);
  hipLaunchKernelGGL((selfattn_fwd_kernel<T, NKS, NDT, QT, SUMROW, PRE, NW>), dim3(tiles * p.H, p.B), dim3(64 * NW), lds, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "selfattn launch: %s", hipGetErrorString(e));
}

template <typename T, int NKS, int NDT, bool SUMROW, bool PRE>
int launch_sa_cfg(const SParams& p, hipStream_t st) {
  constexpr int QT = NDT > 6 ? 1 : 2;     // d > 96: one query tile per wave keeps the 4*NDT accumulator + 8*NDT V^T registers under 256
  // d <= 48 (SD-v1 level 0, the launch that is 10 % of a UNet call): STA_OPT_SELFATTN_WAVES = 8 selects the four-waves-per-SIMD
  // geometry (eight waves x one query tile, 92 registers). Measured SLOWER than four waves x two tiles on the same box (2105 vs
  // 2008 us at 64 x 8 heads, N = 4096: profiles/r04_selfattn_waves.md) — every K / V^T fragment read from LDS then serves 16
  // instead of 32 queries — so it stays an opt-in that the parity tests keep green.
  if constexpr (NDT <= 3 && PRE) {
    if (g_sta_opt[STA_OPT_SELFATTN_WAVES] == 8) return launch_sa_geom<T, NKS, NDT, 1, SUMROW, PRE, 8>(p, st);
    if (g_sta_opt[STA_OPT_SELFATTN_WAVES] == 16) return launch_sa_geom<T, NKS, NDT, 2, SUMROW, PRE, 8>(p, st);      // 256 queries per workgroup
  }
  return launch_sa_geom<T, NKS, NDT, QT, SUMROW, PRE, 4>(p, st);
}

template <typename T, int NKS, int NDT>
int launch_sa(const SParams& p, hipStream_t st) {
  if (p.sl2e == 1.0f) return (p.d & 15) ? launch_sa_cfg<T, NKS, NDT, true, true>(p, st) : launch_sa_cfg<T, NKS, NDT, false, true>(p, st);
  return (p.d & 15) ? launch_sa_cfg<T, NKS, NDT, true, false>(p, st) : launch_sa_cfg<T, NKS, NDT, false, false>(p, st);
}

template <typename T>
int dispatch_sa(const SParams& p, hipStream_t st) {
#ifdef STA_EXPERIMENT_SELFATTN32
  if (p.d > 32 && p.d <= 48 && (p.d & 15) && p.N % 64 == 0 && p.sl2e == 1.0f && g_sta_opt[STA_OPT_SELFATTN_32] != 2) return launch_sa32<T>(p, st);
#endif
  // the level-0 launch (d = 40, 8 heads, q in log2 units, whole 64-key blocks): the software-pipelined loop; STA_OPT_SELFATTN_PIPE = 2
  // keeps the loop above (A/B, parity tests)
  if (p.d == 40 && p.H == 8 && p.sl2e == 1.0f && p.N % KB == 0 && g_sta_opt[STA_OPT_SELFATTN_PIPE] != 2 && g_sta_opt[STA_OPT_SELFATTN_WAVES] != 8)
  {
    // 128 queries per workgroup (four waves, two workgroups per CU); 256 (eight waves, one workgroup per CU) as an option: profiles/r05_selfattn.md
    if (g_sta_opt[STA_OPT_SELFATTN_PIPE] == 8) return launch_sa_pipe<T, 8, 2>(p, st);   // (measured equal to slower than four waves once the padding copies were gone)
    if (g_sta_opt[STA_OPT_SELFATTN_PIPE] == 3) return launch_sa_pipe<T, 4, 3>(p, st);   // three query tiles per wave: 192 queries per workgroup
    if (g_sta_opt[STA_OPT_SELFATTN_PIPE] == 4) return launch_sa_pipe<T, 4, 2>(p, st);
    // Three tiles per wave move a third less through L2 -> LDS per query (-5 % at 64 x 8 heads, N = 4096; -4 % at N = 9216:
    // profiles/r05_selfattn.md) but make 1.5 x longer workgroups: taken when the launch's rounds of 512 resident workgroups (2 per CU)
    // do not quantise worse than with two tiles (8 images x 8 heads at N = 4096: 2.75 rounds of 192-query workgroups against 4.0 of 128)
    const long wg3 = (long)((p.N + 191) / 192) * p.H * p.B, wg2 = (long)((p.N + 127) / 128) * p.H * p.B;
    const long r3 = (wg3 + 511) / 512, r2 = (wg2 + 511) / 512;
    return 57 * r3 < 40 * r2 ? launch_sa_pipe<T, 4, 3>(p, st) : launch_sa_pipe<T, 4, 2>(p, st);      // 3 x 0.95 vs 2 tiles' worth per round
  }
  switch ((p.d + 15) / 16) {
    case 1: return launch_sa<T, 1, 1>(p, st);
    case 2: return launch_sa<T, 1, 2>(p, st);
    case 3: return launch_sa<T, 2, 3>(p, st);
    case 4: return launch_sa<T, 2, 4>(p, st);
    case 5: return launch_sa<T, 3, 5>(p, st);
    case 6: return launch_sa<T, 3, 6>(p, st);
    case 7: return launch_sa<T, 4, 7>(p, st);
    case 8: return launch_sa<T, 4, 8>(p, st);
    case 9: return launch_sa<T, 5, 9>(p, st);
    case 10: return launch_sa<T, 5, 10>(p, st);
  }
  return sta_fail(STA_E_UNSUP, "self-attention head dim %d unsupported (d <= 160)", p.d);
}

}  // namespace

// both 16-bit types at the pipelined kernel's shapes (d = 40, 8 heads, whole 64-key blocks, q in log2 units), that kernel not switched off
extern "C" int sta_selfattn_optimistic_supported(int N, int C, int heads, float scale, int dtype) {
  const float sl2e = scale * 1.4426950408889634f;
  return (dtype == STA_BF16 || dtype == STA_F16) && heads == 8 && C == 320 && N >= KB && N % KB == 0 && fabsf(sl2e - 1.0f) < 1e-6f &&
         g_sta_opt[STA_OPT_SELFATTN_PIPE] != 2 && g_sta_opt[STA_OPT_SELFATTN_WAVES] != 8;
}
// the state words' line (32 words) + one word per workgroup of the widest grid the dispatcher may choose (128 queries per workgroup)
extern "C" size_t sta_selfattn_optimistic_flags_bytes(int B, int N, int heads) {
  return B > 0 && N > 0 && heads > 0 ? ((size_t)B * heads * ((N + 127) / 128) + SA_FLAG0) * sizeof(unsigned) : 0;      // + the state words' 128-byte line
}

static int selfattn_fwd_any(const void* q, const void* k, const void* vt, void* out, float* lse, int B, int N, int C,
                            int heads, int ldq, int ldk, long vt_row_stride, long vt_batch_stride, float scale, int dtype,
                            void* stream, int sfrag = 0, unsigned* flags = nullptr) {
  g_sta_err[0] = 0;
  if (!q || !k || !vt || !out) return sta_fail(STA_E_ARG, "null pointer");
  if (B < 1 || B > 65535 || N < 8 || N % 8 || C <= 0 || heads <= 0 || C % heads)
    return sta_fail(STA_E_ARG, "bad shape B=%d N=%d C=%d heads=%d (need N %% 8 == 0)", B, N, C, heads);
  const int d = C / heads;
  if (d % 8 || d > 160 || ldq < C || ldk < C || ldq % 8 || ldk % 8)
    return sta_fail(STA_E_UNSUP, "self-attention needs d %% 8 == 0, d <= 160, row strides >= C and %% 8 == 0 (d=%d)", d);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  if (vt_row_stride < N || vt_row_stride % 8 || vt_batch_stride % 8)
    return sta_fail(STA_E_ARG, "selfattn: vt strides (%ld, %ld) must be multiples of 8 with row stride >= N", vt_row_stride, vt_batch_stride);
  float sl2e = scale * 1.4426950408889634f;
  if (fabsf(sl2e - 1.0f) < 1e-6f) sl2e = 1.0f;       // scale = ln 2: q is already in log2 units (pre-scaled W_q) -> the PRE kernels
  if (sfrag && !(C == 320 && heads == 8 && N % 16 == 0))
    return sta_fail(STA_E_UNSUP, "self-attention out-fragment order: C = 320 with 8 heads and N %% 16 == 0 (C=%d heads=%d N=%d)", C, heads, N);
  SParams p{q, k, vt, out, B, N, C, heads, d, ldq, ldk, vt_row_stride, vt_batch_stride, sl2e, lse, sfrag, flags, flags ? 1 : 0};
  if (flags && !sta_selfattn_optimistic_supported(N, C, heads, scale, dtype))
    return sta_fail(STA_E_UNSUP, "optimistic self-attention: d = 40 with 8 heads, N %% 64 == 0, q in log2 units (scale = ln 2)");
  hipStream_t st = (hipStream_t)stream;
  return dtype == STA_BF16 ? dispatch_sa<__bf16>(p, st) : dispatch_sa<_Float16>(p, st);
}

extern "C" int sta_selfattn_fwd(const void* q, const void* k, const void* vt, void* out, int B, int N, int C,
                                int heads, int ldq, int ldk, long vt_row_stride, long vt_batch_stride, float scale, int dtype,
                                void* stream) {
  return selfattn_fwd_any(q, k, vt, out, nullptr, B, N, C, heads, ldq, ldk, vt_row_stride, vt_batch_stride, scale, dtype, stream);
}

extern "C" int sta_selfattn_fwd_lse(const void* q, const void* k, const void* vt, void* out, float* lse, int B, int N, int C,
                                    int heads, int ldq, int ldk, long vt_row_stride, long vt_batch_stride, float scale, int dtype,
                                    void* stream) {
  if (!lse) return sta_fail(STA_E_ARG, "null pointer");
  return selfattn_fwd_any(q, k, vt, out, lse, B, N, C, heads, ldq, ldk, vt_row_stride, vt_batch_stride, scale, dtype, stream);
}

extern "C" int sta_selfattn_fwd_sfrag(const void* q, const void* k, const void* vt, void* out_frag, int B, int N, int C,
                                      int heads, int ldq, int ldk, long vt_row_stride, long vt_batch_stride, float scale, int dtype,
                                      void* stream) {
  return selfattn_fwd_any(q, k, vt, out_frag, nullptr, B, N, C, heads, ldq, ldk, vt_row_stride, vt_batch_stride, scale, dtype, stream, 1);
}

extern "C" int sta_selfattn_fwd_optimistic(const void* q, const void* k, const void* vt, void* out, void* flags, int B, int N, int C,
                                           int heads, int ldq, int ldk, long vt_row_stride, long vt_batch_stride, float scale, int dtype,
                                           int sfrag, void* stream) {
  if (!flags) return sta_fail(STA_E_ARG, "null pointer");
  return selfattn_fwd_any(q, k, vt, out, nullptr, B, N, C, heads, ldq, ldk, vt_row_stride, vt_batch_stride, scale, dtype, stream, sfrag ? 1 : 0,
                          (unsigned*)flags);
}
