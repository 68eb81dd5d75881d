// sta_selfattn_bwd.hip — backward of the flash-style self-attention (attn1) for gfx950.
//
// Reference: the autograd of CrossAttention.forward with context = x (ldm/modules/attention.py:175-197, called at
// :274) inside the checkpointed blocks of the weight-optimisation epochs (ldm/modules/diffusionmodules/util.py:123-145,
// ldm/models/diffusion/plms.py:275-277): with S = scale * Q K^T, P = softmax(S), O = P V and an incoming dO,
//   dV = P^T dO,   dP = dO V^T,   dS = P o (dP - delta),  delta_i = sum_c dO_ic O_ic,   dQ = scale dS K,   dK = scale dS^T Q.
// Nothing of size N x N is kept by the forward: it leaves the log-sum-exp of every score row (sta_selfattn_fwd_lse) and
// both kernels here rebuild P = exp2(S * log2e - lse) tile by tile, so no running maximum, no row reduction and no
// rescale appear in the loops.
//
// Two kernels, each shaped like the forward (sta_selfattn.hip: 16x16x32 MFMA, 64-row blocks of the streamed side copied
// to LDS once per workgroup by LDS-DMA, double-buffered, one barrier per block), no atomics, deterministic:
//   dq kernel   query pixel = MFMA column (resident Q, dO, lse, delta per lane); streams 64-key blocks of K, V rows and
//               K^T:   S^T = K Q^T,  dP^T = V dO^T,  dS^T = P^T o (dP^T - delta)  -> B operand,  dQ^T += K^T dS^T
//   dkv kernel  key = MFMA column (resident K, V); streams 64-pixel blocks of Q, dO rows, Q^T, dO^T and (lse, delta):
//               S = Q K^T,  dP = dO V^T,  P -> B operand: dV^T += dO^T P;  dS -> B operand: dK^T += Q^T dS
// Rows of an S tile are assigned to the streamed index exactly as in the forward (tile_key), so P and dS leave the VALU
// already in B-operand order with 8 consecutive streamed rows per lane, and the transposed copies (K^T, Q^T, dO^T:
// [B][C][N], made by the caller with one transpose each) are read 16 B per lane.
// The streamed side must be a whole number of 64-row blocks (N % 64 == 0: every level of the SD-v1 UNet).

#include "sta_selfattn_dev.h"

namespace {

struct SBParams {
  const void *q, *k, *v;        // [B][N][ld] rows; head h at column h*d (slices of one fused [B][N][3C] buffer are fine)
  const void *qt, *kt, *dot;    // [B][C][N] transposed copies of q, k, dout
  const void *dout, *out;       // [B][N][C]
  const float* lse;             // [B][H][N], log2 domain (sta_selfattn_fwd_lse)
  float* delta;                 // [B][H][N] workspace
  void *dq, *dk, *dv;           // [B][N][ldg] rows
  int B, N, C, H, d, ld, ldg;
  float sl2e, scale;
};

extern __shared__ __attribute__((aligned(16))) char smem_sb[];

// delta[b][h][n] = sum over the head's channels of dout * out (fp32)
template <typename T>
__global__ __launch_bounds__(256) void selfattn_delta_kernel(const SBParams p) {
  using V8 = typename Tr<T>::V8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)p.B * p.N * p.H;
  if (i >= total) return;
  const int h = (int)(i % p.H);
  const long bn = i / p.H;
  const int n = (int)(bn % p.N), b = (int)(bn / p.N);
  const T* a = (const T*)p.dout + bn * p.C + h * p.d;
  const T* o = (const T*)p.out + bn * p.C + h * p.d;
  float s = 0.f;
  for (int c = 0; c < p.d; c += 8) {
    const V8 x = *(const V8*)(a + c), y = *(const V8*)(o + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)x[j] * (float)y[j];
  }
  p.delta[((size_t)b * p.H + h) * p.N + n] = s;
}

// PRE (q in log2 units: scale * log2 e == 1, what the model's pre-scaled W_q gives): S - lse and dP - delta come out of the MFMAs — the
// accumulators START at -lse / -delta (per query: a lane scalar) — so the vector work per score is exp2, one multiply and the
// conversion instead of fma, exp2, subtract, multiply, conversion (matrix and vector cycles add up in these loops: SQ counters, 45 % + 49 %).
template <typename T, int NKS, int NDT, int QT, int NW = 4, bool PRE = false>
__global__ __launch_bounds__(64 * NW) void selfattn_bwd_dq_kernel(const SBParams p) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  constexpr int NKF = 4 * NKS;            // K (and V) row fragments per block: 4 key tiles x NKS head-dim steps
  constexpr int NTF = 2 * NDT;            // K^T fragments per block: 2 key steps x NDT head-dim tiles
  constexpr int NFR = 2 * NKF + NTF;      // [K | V | K^T]
  constexpr int PER = (NFR + NW - 1) / NW;
  constexpr int BB = NW * PER * FRAG;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int N = p.N, d = p.d;
  const int b = blockIdx.y;
  const int tile = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int px0 = (tile * NW + wv) * 16 * QT;

  const T* qb = (const T*)p.q + (size_t)b * N * p.ld + h * d;
  const T* kb = (const T*)p.k + (size_t)b * N * p.ld + h * d;
  const T* vb = (const T*)p.v + (size_t)b * N * p.ld + h * d;
  const T* gb = (const T*)p.dout + (size_t)b * N * p.C + h * d;
  const T* ktb = (const T*)p.kt + ((size_t)b * p.C + h * d) * N;
  const float* Lb = p.lse + ((size_t)b * p.H + h) * N;
  const float* Db = p.delta + ((size_t)b * p.H + h) * N;

  // resident B operands (zero in the padded head-dim slots, so the clamped K / V columns never matter) and row scalars
  V8 qf[QT][NKS], gf[QT][NKS];
  float L[QT], dl[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int px = px0 + 16 * qt + c16;
    const bool in = px < N;
    L[qt] = in ? Lb[px] : 0.f;
    dl[qt] = in ? Db[px] : 0.f;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      V8 z = {};
      const int dd = 32 * s + 8 * g;
      qf[qt][s] = (in && dd < d) ? *(const V8*)(qb + (size_t)px * p.ld + dd) : z;
      gf[qt][s] = (in && dd < d) ? *(const V8*)(gb + (size_t)px * p.C + dd) : z;
    }
  }

  const char* src[PER];
  int step[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int f = wv + NW * i;
    const int fs = f < NFR ? f : 0;
    if (fs < 2 * NKF) {
      const int f1 = fs < NKF ? fs : fs - NKF;
      const int Tt = f1 / NKS, s = f1 - Tt * NKS;
      src[i] = (const char*)((fs < NKF ? kb : vb) + (size_t)tile_key(Tt, c16) * p.ld + min(32 * s + 8 * g, d - 8));
      step[i] = __builtin_amdgcn_readfirstlane(KB * p.ld * (int)sizeof(T));
    } else {
      const int f2 = fs - 2 * NKF;
      const int s2 = f2 / NDT, u = f2 - s2 * NDT;
      src[i] = (const char*)(ktb + (size_t)min(16 * u + c16, d - 1) * N + 32 * s2 + 8 * g);
      step[i] = __builtin_amdgcn_readfirstlane(KB * (int)sizeof(T));
    }
  }
  auto stage = [&](char* dst) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                       (__attribute__((address_space(3))) void*)(dst + (wv + NW * i) * FRAG), 16, 0, 0);
      src[i] += step[i];
    }
  };

  f32x4 dq[QT][NDT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int u = 0; u < NDT; ++u) dq[qt][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nblk = N / KB;
  stage(smem_sb);
  for (int blk = 0; blk < nblk; ++blk) {
    char* cur = smem_sb + (blk & 1) * BB;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // block `blk` landed for every wave; everyone left block blk-1
    if (blk + 1 < nblk) stage(smem_sb + ((blk + 1) & 1) * BB);
    const V8* fr = (const V8*)cur + lane;
    f32x4 st[QT][4], dp[QT][4];
    {
      V8 ka[NKF];
#pragma unroll
      for (int f = 0; f < NKF; ++f) ka[f] = fr[f * 64];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float i0 = PRE ? -L[qt] : 0.f;
          st[qt][t] = f32x4{i0, i0, i0, i0};
#pragma unroll
          for (int s = 0; s < NKS; ++s) st[qt][t] = Tr<T>::mfma(ka[t * NKS + s], qf[qt][s], st[qt][t]);
        }
      __builtin_amdgcn_s_setprio(0);
    }
    {
      V8 va[NKF];
#pragma unroll
      for (int f = 0; f < NKF; ++f) va[f] = fr[(NKF + f) * 64];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float i0 = PRE ? -dl[qt] : 0.f;
          dp[qt][t] = f32x4{i0, i0, i0, i0};
#pragma unroll
          for (int s = 0; s < NKS; ++s) dp[qt][t] = Tr<T>::mfma(va[t * NKS + s], gf[qt][s], dp[qt][t]);
        }
      __builtin_amdgcn_s_setprio(0);
    }
    V8 ta[NTF];
#pragma unroll
    for (int f = 0; f < NTF; ++f) ta[f] = fr[(2 * NKF + f) * 64];
    V8 dsb[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (PRE) {
            st[qt][t][r] = __builtin_amdgcn_exp2f(st[qt][t][r]) * dp[qt][t][r];
          } else {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qt][t][r], p.sl2e, -L[qt]));
            st[qt][t][r] = e * (dp[qt][t][r] - dl[qt]);
          }
        }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) dsb[qt][s2][j] = (T)st[qt][2 * s2 + (j >> 2)][j & 3];
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < NDT; ++u)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) dq[qt][u] = Tr<T>::mfma(ta[s2 * NDT + u], dsb[qt][s2], dq[qt][u]);
    __builtin_amdgcn_s_setprio(0);
  }

#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int px = px0 + 16 * qt + c16;
    if (px >= N) continue;
    T* ob = (T*)p.dq + ((size_t)b * N + px) * p.ldg + h * d;
#pragma unroll
    for (int u = 0; u < NDT; ++u) {
      const int dd = 16 * u + 4 * g;
      if (dd < d) {
        V4 r4;
#pragma unroll
        for (int r = 0; r < 4; ++r) r4[r] = (T)(dq[qt][u][r] * p.scale);
        *(V4*)(ob + dd) = r4;
      }
    }
  }
}

// (The dq kernel's accumulator-start trick was built here too — resident K / V negated once, accumulators starting at +lse / +delta read as row
// scalars from LDS, 19 % fewer vector instructions in the loop — and measured equal, as did operand reads issued a phase ahead of their MFMAs
// and eight waves sharing a block in situ: this loop is 45 % matrix-pipe busy and answers to none of the three. Left in the plain form.)
template <typename T, int NKS, int NDT, int KT_, int NW = 4>
__global__ __launch_bounds__(64 * NW) void selfattn_bwd_dkv_kernel(const SBParams p) {
  using V8 = typename Tr<T>::V8;
  using V4 = typename Tr<T>::V4;
  constexpr int NQF = 4 * NKS;            // Q (and dO) row fragments per block: 4 pixel tiles x NKS head-dim steps
  constexpr int NTF = 2 * NDT;            // Q^T (and dO^T) fragments per block: 2 pixel steps x NDT head-dim tiles
  constexpr int NFR = 2 * NQF + 2 * NTF + 1;   // [Q | dO | Q^T | dO^T | (lse, delta)]
  constexpr int PER = (NFR + NW - 1) / NW;
  constexpr int BB = NW * PER * FRAG;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int N = p.N, d = p.d;
  const int b = blockIdx.y;
  const int tile = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int key0 = (tile * NW + wv) * 16 * KT_;

  const T* qb = (const T*)p.q + (size_t)b * N * p.ld + h * d;
  const T* kb = (const T*)p.k + (size_t)b * N * p.ld + h * d;
  const T* vb = (const T*)p.v + (size_t)b * N * p.ld + h * d;
  const T* gb = (const T*)p.dout + (size_t)b * N * p.C + h * d;
  const T* qtb = (const T*)p.qt + ((size_t)b * p.C + h * d) * N;
  const T* gtb = (const T*)p.dot + ((size_t)b * p.C + h * d) * N;
  const float* Lb = p.lse + ((size_t)b * p.H + h) * N;
  const float* Db = p.delta + ((size_t)b * p.H + h) * N;

  V8 kf[KT_][NKS], vf[KT_][NKS];
#pragma unroll
  for (int kt = 0; kt < KT_; ++kt) {
    const int key = key0 + 16 * kt + c16;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      V8 z = {};
      const int dd = 32 * s + 8 * g;
      const bool in = key < N && dd < d;
      kf[kt][s] = in ? *(const V8*)(kb + (size_t)key * p.ld + dd) : z;
      vf[kt][s] = in ? *(const V8*)(vb + (size_t)key * p.ld + dd) : z;
    }
  }

  const char* src[PER];
  int step[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int f = wv + NW * i;
    const int fs = f < NFR ? f : 0;
    if (fs < 2 * NQF) {
      const int f1 = fs < NQF ? fs : fs - NQF;
      const int Tt = f1 / NKS, s = f1 - Tt * NKS;
      src[i] = fs < NQF ? (const char*)(qb + (size_t)tile_key(Tt, c16) * p.ld + min(32 * s + 8 * g, d - 8))
                        : (const char*)(gb + (size_t)tile_key(Tt, c16) * p.C + min(32 * s + 8 * g, d - 8));
      step[i] = __builtin_amdgcn_readfirstlane(KB * (fs < NQF ? p.ld : p.C) * (int)sizeof(T));
    } else if (fs < 2 * NQF + 2 * NTF) {
      const int f2 = fs - 2 * NQF;
      const int f3 = f2 < NTF ? f2 : f2 - NTF;
      const int s2 = f3 / NDT, u = f3 - s2 * NDT;
      src[i] = (const char*)((f2 < NTF ? qtb : gtb) + (size_t)min(16 * u + c16, d - 1) * N + 32 * s2 + 8 * g);
      step[i] = __builtin_amdgcn_readfirstlane(KB * (int)sizeof(T));
    } else {       // lanes 0..15: lse of pixels 4*lane..+3; lanes 16..31: delta; the upper half repeats them
      src[i] = (const char*)(((lane & 16) ? Db : Lb) + 4 * c16);
      step[i] = __builtin_amdgcn_readfirstlane(KB * (int)sizeof(float));
    }
  }
  auto stage = [&](char* dst) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                       (__attribute__((address_space(3))) void*)(dst + (wv + NW * i) * FRAG), 16, 0, 0);
      src[i] += step[i];
    }
  };

  f32x4 dk[KT_][NDT], dv[KT_][NDT];
#pragma unroll
  for (int kt = 0; kt < KT_; ++kt)
#pragma unroll
    for (int u = 0; u < NDT; ++u) dk[kt][u] = dv[kt][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Two LDS buffers where they fit (d <= 96); above, one 64-pixel block of [Q | dO | Q^T | dO^T] is 81 KB at d = 160 and the
  // loop is stage -> barrier -> compute -> barrier (those levels have N <= 576: at most 9 blocks).
  constexpr bool DB = 2 * BB <= 160 * 1024;
  const int nblk = N / KB;
  if constexpr (DB) stage(smem_sb);
  for (int blk = 0; blk < nblk; ++blk) {
    char* cur = DB ? smem_sb + (blk & 1) * BB : smem_sb;
    if constexpr (!DB) {
      if (blk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // every wave has read block blk-1 out of the single buffer
      }
      stage(cur);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DB) {
      if (blk + 1 < nblk) stage(smem_sb + ((blk + 1) & 1) * BB);
    }
    const V8* fr = (const V8*)cur + lane;
    f32x4 st[KT_][4], dp[KT_][4];
    // row scalars of this lane's 16 pixels: tile t, rows 4g..4g+3 <-> 4 consecutive pixels tile_key(t, 4g) ..
    const float* ld_ = (const float*)(cur + (2 * NQF + 2 * NTF) * FRAG);
    f32x4 L4[4], D4[4];
    {
      V8 qa[NQF];
#pragma unroll
      for (int f = 0; f < NQF; ++f) qa[f] = fr[f * 64];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kt = 0; kt < KT_; ++kt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          st[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < NKS; ++s) st[kt][t] = Tr<T>::mfma(qa[t * NKS + s], kf[kt][s], st[kt][t]);
        }
      __builtin_amdgcn_s_setprio(0);
    }
    {
      V8 ga[NQF];
#pragma unroll
      for (int f = 0; f < NQF; ++f) ga[f] = fr[(NQF + f) * 64];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kt = 0; kt < KT_; ++kt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          dp[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < NKS; ++s) dp[kt][t] = Tr<T>::mfma(ga[t * NKS + s], vf[kt][s], dp[kt][t]);
        }
      __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      L4[t] = *(const f32x4*)(ld_ + tile_key(t, 4 * g));
      D4[t] = *(const f32x4*)(ld_ + 64 + tile_key(t, 4 * g));
    }
    V8 pb[KT_][2], dsb[KT_][2];
#pragma unroll
    for (int kt = 0; kt < KT_; ++kt) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][t][r], p.sl2e, -L4[t][r]));
          st[kt][t][r] = e;
          dp[kt][t][r] = e * (dp[kt][t][r] - D4[t][r]);
        }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pb[kt][s2][j] = (T)st[kt][2 * s2 + (j >> 2)][j & 3];
          dsb[kt][s2][j] = (T)dp[kt][2 * s2 + (j >> 2)][j & 3];
        }
    }
    {
      V8 ta[NTF];
#pragma unroll
      for (int f = 0; f < NTF; ++f) ta[f] = fr[(2 * NQF + NTF + f) * 64];      // dO^T
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int u = 0; u < NDT; ++u)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int kt = 0; kt < KT_; ++kt) dv[kt][u] = Tr<T>::mfma(ta[s2 * NDT + u], pb[kt][s2], dv[kt][u]);
      __builtin_amdgcn_s_setprio(0);
    }
    {
      V8 ta[NTF];
#pragma unroll
      for (int f = 0; f < NTF; ++f) ta[f] = fr[(2 * NQF + f) * 64];            // Q^T
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int u = 0; u < NDT; ++u)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int kt = 0; kt < KT_; ++kt) dk[kt][u] = Tr<T>::mfma(ta[s2 * NDT + u], dsb[kt][s2], dk[kt][u]);
      __builtin_amdgcn_s_setprio(0);
    }
  }

#pragma unroll
  for (int kt = 0; kt < KT_; ++kt) {
    const int key = key0 + 16 * kt + c16;
    if (key >= N) continue;
    T* okb = (T*)p.dk + ((size_t)b * N + key) * p.ldg + h * d;
    T* ovb = (T*)p.dv + ((size_t)b * N + key) * p.ldg + h * d;
#pragma unroll
    for (int u = 0; u < NDT; ++u) {
      const int dd = 16 * u + 4 * g;
      if (dd < d) {
        V4 a4, b4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a4[r] = (T)(dk[kt][u][r] * p.scale);
          b4[r] = (T)dv[kt][u][r];
        }
        *(V4*)(okb + dd) = a4;
        *(V4*)(ovb + dd) = b4;
      }
    }
  }
}

template <typename T, int NKS, int NDT, int QT, int NW = 4, bool PRE = false>
int launch_sb(const SBParams& p, hipStream_t st) {
  constexpr int lds_q = 2 * NW * ((2 * 4 * NKS + 2 * NDT + NW - 1) / NW) * FRAG;
  constexpr int bb_kv = NW * ((2 * 4 * NKS + 4 * NDT + 1 + NW - 1) / NW) * FRAG;
  constexpr int lds_kv = (2 * bb_kv <= 160 * 1024 ? 2 : 1) * bb_kv;
  static StaLdsAttr attr_q, attr_kv;
  if (!attr_q.ensure((const void*)selfattn_bwd_dq_kernel<T, NKS, NDT, QT, NW, PRE>, lds_q) ||
      !attr_kv.ensure((const void*)selfattn_bwd_dkv_kernel<T, NKS, NDT, QT, NW>, lds_kv))
    return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(selfattn backward) failed");
  const long total = (long)p.B * p.N * p.H;
  hipLaunchKernelGGL((selfattn_delta_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
  const int tiles = (p.N + 16 * NW * QT - 1) / (16 * NW * QT);
  hipLaunchKernelGGL((selfattn_bwd_dkv_kernel<T, NKS, NDT, QT, NW>), dim3(tiles * p.H, p.B), dim3(64 * NW), lds_kv, st, p);
  hipLaunchKernelGGL((selfattn_bwd_dq_kernel<T, NKS, NDT, QT, NW, PRE>), dim3(tiles * p.H, p.B), dim3(64 * NW), lds_q, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "selfattn backward launch: %s", hipGetErrorString(e));
}

template <typename T>
int dispatch_sb(const SBParams& p, hipStream_t st) {
  // q in log2 units (scale = ln 2: the model's attn1 with its pre-scaled W_q): the PRE kernels, built for the head dims of SD-v1 (40, 80, 160)
  const bool pre = p.sl2e == 1.0f;
  switch ((p.d + 15) / 16) {
    case 1: return launch_sb<T, 1, 1, 2>(p, st);
    case 2: return launch_sb<T, 1, 2, 2>(p, st);
    case 3:
      // d = 40 (SD-v1 level 0: N = 4096): eight waves share a streamed 64-row block where four did (22 / 29 KiB per block and workgroup
      // through L2 -> LDS) — same waves per SIMD, half the stream; -4 % stand-alone. STA_OPT_SELFATTN_WAVES = 4 keeps four.
      if (p.N >= 1024 && g_sta_opt[STA_OPT_SELFATTN_WAVES] != 4) return pre ? launch_sb<T, 2, 3, 2, 8, true>(p, st) : launch_sb<T, 2, 3, 2, 8>(p, st);
      return pre ? launch_sb<T, 2, 3, 2, 4, true>(p, st) : launch_sb<T, 2, 3, 2>(p, st);
    case 4: return launch_sb<T, 2, 4, 2>(p, st);
    case 5: return pre ? launch_sb<T, 3, 5, 1, 4, true>(p, st) : launch_sb<T, 3, 5, 1>(p, st);
    case 6: return launch_sb<T, 3, 6, 1>(p, st);
    case 7: return launch_sb<T, 4, 7, 1>(p, st);
    case 8: return launch_sb<T, 4, 8, 1>(p, st);
    case 9: return launch_sb<T, 5, 9, 1>(p, st);
    case 10: return pre ? launch_sb<T, 5, 10, 1, 4, true>(p, st) : launch_sb<T, 5, 10, 1>(p, st);
  }
  return sta_fail(STA_E_UNSUP, "self-attention backward: head dim %d unsupported (d <= 160)", p.d);
}

}  // namespace

extern "C" int sta_selfattn_bwd(const void* q, const void* k, const void* v, const void* qt, const void* kt, const void* dout,
                                const void* doutt, const void* out, const float* lse, float* delta, void* dq, void* dk, void* dv,
                                int B, int N, int C, int heads, int ld, int ldg, float scale, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!q || !k || !v || !qt || !kt || !dout || !doutt || !out || !lse || !delta || !dq || !dk || !dv)
    return sta_fail(STA_E_ARG, "null pointer");
  if (B < 1 || B > 65535 || N < 64 || C <= 0 || heads <= 0 || C % heads)
    return sta_fail(STA_E_ARG, "bad shape B=%d N=%d C=%d heads=%d", B, N, C, heads);
  const int d = C / heads;
  if (N % 64 || d % 8 || d > 160 || ld < C || ldg < C || ld % 8 || ldg % 4)
    return sta_fail(STA_E_UNSUP, "self-attention backward needs N %% 64 == 0, d %% 8 == 0, d <= 160, row strides >= C (N=%d d=%d)", N, d);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  float sl2e = scale * 1.4426950408889634f;
  if (fabsf(sl2e - 1.0f) < 1e-6f) sl2e = 1.0f;       // scale = ln 2: q in log2 units — the same snap as the forward that wrote `lse`
  SBParams p{q, k, v, qt, kt, doutt, dout, out, lse, delta, dq, dk, dv, B, N, C, heads, d, ld, ldg, sl2e, scale};
  hipStream_t st = (hipStream_t)stream;
  return dtype == STA_BF16 ? dispatch_sb<__bf16>(p, st) : dispatch_sb<_Float16>(p, st);
}
