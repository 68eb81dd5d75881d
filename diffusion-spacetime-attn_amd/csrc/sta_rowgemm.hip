// sta_rowgemm.hip — the tail of the fused cross-attention chain at SD-v1 level 0 (C = 320, 8 heads of 40):
//
//     s = x + to_out(blended) ;  y = LayerNorm(s)                      (attention.py:215 to_out, :294-299 residual + norm3)
//
// in ONE pass, reading `blended` in the OUT-FRAGMENT order the head-pair attention kernel writes (sta_xattn_proj3.hip, OF mode:
// sta_p3::ofrag_channel). Row-major, the same work is a library GEMM (write 168 MB of to_out(blended), 32 images) followed by
// the add + LayerNorm pass (read it back): this kernel never materialises to_out's result. SURVEY.md section 8f rank 1, the
// `to_out` half: with sta_add_layernorm_qfrag -> sta_xattn_fwd_proj_qfrag_ofrag -> this kernel neither norm2's output, nor q,
// nor the blended pre-projection tensor, nor to_out's output exists in row-major form in HBM.
//
// Shape of the computation. Rows are the MFMA COLUMNS (as everywhere in this library): Out^T [320 x 16 rows] = Wo' . A^T with
//   B operand = one 1-KiB fragment of the blended tensor per k-step (lane (g, c): 8 channels of row c) — a plain coalesced load,
//   A operand = to_out.weight re-laid out once per model (pack_wo): fragment (row tile u, k-step f) in lane order, its k-slots
//               following ofrag_channel(f, g, j) and its rows sigma(u, rho) = 32 (u >> 1) + 8 (rho >> 2) + 4 (u & 1) + (rho & 3),
//               so that the accumulators of row tiles 2v | 2v + 1 in lane (g, c) are the 8 CONSECUTIVE output channels
//               32 v + 8 g .. + 7 of row c: the residual load, both stores and the gamma / beta reads are 16 bytes per lane.
// The weight is 200 KiB of fragments — more than a CU's LDS — and every row needs all of it: it is STREAMED, 10 chunks of two
// row tiles (20 KiB) per pass of 128 rows (8 waves x 16), through a 2-slot LDS ring fed by LDS-DMA one chunk ahead (L2-resident:
// 205 MB of L2 -> LDS traffic per 262144 rows, beside 672 MB of HBM traffic that bounds the pass). A wave keeps its 16 rows'
// ten B fragments in registers (40) and all 20 accumulator tiles (80) — the whole output row of its pixels, which is what lets
// the LayerNorm statistics stay inside the wave (80 values per lane, 4 lanes per row: two permlane swaps).
//
// Roofline: HBM. Algorithmic bytes per row: 640 (blended) + 640 (x) + 640 (s) + 640 (y) = 2560 B; 2 * 320 * 320 flop = 80 flop/B.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sta_xattn.h"
#include "sta_internal.h"
#include "sta_xattn_dev.h"
#include "sta_xattn_proj3.h"

namespace {

constexpr int RG_C = 320;                  // channels (in = out)
constexpr int RG_NKS = RG_C / 32;          // 10 k-steps
constexpr int RG_NRT = RG_C / 16;          // 20 output row tiles
constexpr int RG_NCH = RG_NRT / 2;         // 10 chunks of two row tiles
constexpr int RG_NW = 8;                   // waves per workgroup
constexpr int RG_CHUNK_FR = 2 * RG_NKS;    // 20 fragments per chunk
constexpr int RG_PER = (RG_CHUNK_FR + RG_NW - 1) / RG_NW;   // LDS-DMA instructions per wave per chunk (3; 4 padding copies)
constexpr int RG_SLOT = RG_PER * RG_NW * FRAG;              // 24 KiB per ring slot
constexpr int RG_TAB = 3 * RG_C * 2;       // bias | gamma | beta as 16-bit, behind the ring
constexpr int RG_LDS = 2 * RG_SLOT + RG_TAB;

__host__ __device__ constexpr int rg_sigma(int u, int rho) { return 32 * (u >> 1) + 8 * (rho >> 2) + 4 * (u & 1) + (rho & 3); }

// Channel behind slot j of lane row g of fragment f of the SELF-attention kernel's out-fragment order (sta_selfattn_fwd_sfrag):
// f < 8: head f, O^T rows 4g + j of tile 0 | 16 + 4g + (j - 4) of tile 1 (= head dims, no permutation there);
// f >= 8: lane row g is head 4 (f - 8) + g, slots = its dims 32 .. 39.
__host__ __device__ constexpr int sfrag_channel(int f, int g, int j) {
  return f < 8 ? f * 40 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)) : (4 * (f - 8) + g) * 40 + 32 + j;
}

// to_out.weight [C out][C in] -> [chunk v][t][k-step f] fragments: lane (g, c) of fragment (u = 2v + t, f) holds
// Wo[sigma(u, c)][channel(f, g, 0 .. 7)], channel = ofrag_channel (kind 0: the cross-attention kernel's order) or sfrag_channel (kind 1)
template <typename T>
__global__ __launch_bounds__(64) void pack_wo_ofrag_kernel(const T* __restrict__ wo, T* __restrict__ packed, int kind) {
  const int fr = blockIdx.x;               // (v * 2 + t) * NKS + f
  const int u = fr / RG_NKS, f = fr % RG_NKS;
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int row = rg_sigma(u, c);
  T* dst = packed + (size_t)fr * (FRAG / 2) + lane * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j] = wo[(size_t)row * RG_C + (kind ? sfrag_channel(f, g, j) : sta_p3::ofrag_channel(f, g, j))];
}

struct RG {
  const char* a;        // blended, out-fragment order [R / 16][10][1 KiB]
  const char* wo;       // packed weight
  const void* bias;     // [C] or null
  const void* gamma;
  const void* beta;
  const void* x;        // [R][C] residual stream
  void* s;              // [R][C] new residual stream (x + to_out(blended) + bias)
  void* y;              // [R][C] LayerNorm(s)
  long R;
  float eps;
  int y_qfrag;          // 1: y leaves in QUERY-fragment order (it is norm2's output feeding sta_xattn_fwd_proj_qfrag*)
};

template <typename T>
__global__ __launch_bounds__(64 * RG_NW, 2) void to_out_ln_ofrag_kernel(const RG p) {
  using V8 = typename Tr<T>::V8;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  char* ring = smem;
  T* tab = (T*)(smem + 2 * RG_SLOT);
  // tables: bias | gamma | beta (bias may be absent: zeros)
  for (int i = threadIdx.x; i < RG_C; i += 64 * RG_NW) {
    tab[i] = p.bias ? ((const T*)p.bias)[i] : (T)0.0f;
    tab[RG_C + i] = ((const T*)p.gamma)[i];
    tab[2 * RG_C + i] = ((const T*)p.beta)[i];
  }
  const long nblk = (p.R + 16 * RG_NW - 1) / (16 * RG_NW);
  // LDS-DMA of weight chunk `ch` into ring slot `slot`: 20 fragments, waves round-robin, the 4 spare positions re-read fragment 0
  // (buffer form: the per-lane part of the address is ONE register, lane * 16; the fragment is a scalar offset)
  const __amdgpu_buffer_rsrc_t w_srd = make_srd(p.wo, (unsigned)(RG_NRT * RG_NKS * FRAG));
  const unsigned lane16 = (unsigned)lane * 16u;
  auto stage = [&](int ch, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RG_PER; ++i) {
      const int f = wv + RG_NW * i;
      const int fs = f < RG_CHUNK_FR ? f : 0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_srd, (__attribute__((address_space(3))) void*)(ring + slot * RG_SLOT + f * FRAG), 16, lane16,
                                           (unsigned)((ch * RG_CHUNK_FR + fs) * FRAG), 0, 0);
    }
  };
  const size_t total = (size_t)p.R * RG_C * sizeof(T);
  const __amdgpu_buffer_rsrc_t a_srd = make_srd(p.a, (unsigned)(total > 0xfffffff0ull ? 0xfffffff0ull : total));
  // (activations above 4 GiB are refused by the host wrapper: 32-bit buffer offsets)
  auto a_off = [&](long blk) -> unsigned {
    const long row0 = (blk * RG_NW + wv) * 16;
    return (blk < nblk && row0 < p.R) ? (unsigned)(row0 * RG_C * (long)sizeof(T)) + (unsigned)lane * 16u : 0xfffffff0u;
  };
  long blk = blockIdx.x;
  V8 b[RG_NKS];
  stage(0, 0);
  {
    const unsigned vo = a_off(blk);
#pragma unroll
    for (int f = 0; f < RG_NKS; ++f) b[f] = srd_load16<V8>(a_srd, vo, 1024u * f);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                         // tables + chunk 0 visible
  const char* lbase = ring + lane * 16;
  for (; blk < nblk; blk += gridDim.x) {
    f32x4 acc[RG_NRT];
    auto chunk = [&](auto ch_tag) __attribute__((always_inline)) {
      constexpr int CH = decltype(ch_tag)::value;
      constexpr int SLOT = CH & 1;
      // VMEM operations younger than this chunk's DMA that may stay in flight: the previous pass's 20 stores (CH = 0: they follow
      // the DMA of "chunk 0 of the next pass", issued during chunk 9); otherwise the DMA is the newest operation
      if constexpr (CH == 0) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();        // chunk CH landed for every wave; everyone is done with the other slot
      stage((CH + 1) % RG_NCH, SLOT ^ 1);  // the next chunk (chunk 0 of the next pass behind chunk 9)
      // all 20 operand fragments of the chunk requested up front (80 registers), the MFMAs follow them in request order: one
      // LDS latency per chunk instead of one per MFMA (hipcc otherwise sinks every read to its use)
      const V8* fr = (const V8*)(lbase + SLOT * RG_SLOT);
      V8 wa[2][RG_NKS];
#pragma unroll
      for (int f = 0; f < RG_NKS; ++f) {
        wa[0][f] = fr[f * 64];
        wa[1][f] = fr[(RG_NKS + f) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int f = 0; f < RG_NKS; ++f) {
        a0 = Tr<T>::mfma(wa[0][f], b[f], a0);
        a1 = Tr<T>::mfma(wa[1][f], b[f], a1);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[2 * CH] = a0;
      acc[2 * CH + 1] = a1;
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{});
    chunk(std::integral_constant<int, 3>{}); chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{});
    chunk(std::integral_constant<int, 6>{}); chunk(std::integral_constant<int, 7>{}); chunk(std::integral_constant<int, 8>{});
    chunk(std::integral_constant<int, 9>{});
    {   // the NEXT item's B fragments into the registers the last MFMA just released: they land under the epilogue
      const unsigned vo = a_off(blk + gridDim.x);
#pragma unroll
      for (int f = 0; f < RG_NKS; ++f) b[f] = srd_load16<V8>(a_srd, vo, 1024u * f);
    }
    // ---- epilogue: + bias + residual, new residual stream, LayerNorm ------------------------------------------------
    const long row0 = (blk * RG_NW + wv) * 16;      // wave-uniform; R % 16 == 0: a wave's 16 rows exist together or not at all
    if (row0 < p.R) {
    const long row = row0 + c16;
    const T* xr = (const T*)p.x + row * RG_C + 8 * g;
    V8 xv[RG_NCH];
#pragma unroll
    for (int v = 0; v < RG_NCH; ++v) xv[v] = *(const V8*)(xr + 32 * v);
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < RG_NCH; ++v) {
      const V8 bs = *(const V8*)(tab + 32 * v + 8 * g);
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = acc[2 * v + (e >> 2)][e & 3] + (float)bs[e] + (float)xv[v][e];
        o[e] = (T)t;                       // the residual stream continues in the activation dtype: normalise what is stored
        const float r = (float)o[e];
        acc[2 * v + (e >> 2)][e & 3] = r;
        sum += r;
      }
      *(V8*)((T*)p.s + row * RG_C + 32 * v + 8 * g) = o;
      __builtin_amdgcn_sched_barrier(0);   // one tile pair at a time: hipcc otherwise hoists all 30 table reads (120 registers) to the top
    }
    const float mean = bfly_sum(sum) * (1.0f / RG_C);
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < RG_NRT; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[u][r] - mean;
        q += d * d;
      }
    const float rstd = rsqrtf(bfly_sum(q) * (1.0f / RG_C) + p.eps);
#pragma unroll
    for (int v = 0; v < RG_NCH; ++v) {
      const V8 gm = *(const V8*)(tab + RG_C + 32 * v + 8 * g), bt = *(const V8*)(tab + 2 * RG_C + 32 * v + 8 * g);
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (T)((acc[2 * v + (e >> 2)][e & 3] - mean) * rstd * (float)gm[e] + (float)bt[e]);
      // query-fragment order: tile pair v of lane (g, c) IS fragment v's lane 16 g + c (channels 32 v + 8 g .. + 7 of row c)
      if (p.y_qfrag) *(V8*)((char*)p.y + (size_t)row0 * RG_C * sizeof(T) + v * FRAG + lane * 16) = o;
      else *(V8*)((T*)p.y + row * RG_C + 32 * v + 8 * g) = o;
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing DMA of "chunk 0 of the next pass" must not outlive the workgroup
}

}  // namespace

extern "C" {

size_t sta_to_out_ln_packed_wo_bytes(int C, int heads) {
  return (C == RG_C && heads == 8) ? (size_t)RG_NRT * RG_NKS * FRAG : 0;
}

int sta_to_out_ln_pack_wo(const void* wo, void* packed, int C, int heads, int kind, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!wo || !packed) return sta_fail(STA_E_ARG, "null pointer");
  if (sta_to_out_ln_packed_wo_bytes(C, heads) == 0) return sta_fail(STA_E_UNSUP, "to_out + LayerNorm in out-fragment order: C = 320 with 8 heads only (C=%d heads=%d)", C, heads);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  if (kind != 0 && kind != 1) return sta_fail(STA_E_ARG, "fragment kind %d (0: cross-attention out fragments, 1: self-attention out fragments)", kind);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == STA_BF16) hipLaunchKernelGGL(pack_wo_ofrag_kernel<__bf16>, dim3(RG_NRT * RG_NKS), dim3(64), 0, st, (const __bf16*)wo, (__bf16*)packed, kind);
  else hipLaunchKernelGGL(pack_wo_ofrag_kernel<_Float16>, dim3(RG_NRT * RG_NKS), dim3(64), 0, st, (const _Float16*)wo, (_Float16*)packed, kind);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "pack_wo_ofrag launch: %s", hipGetErrorString(e));
}

int sta_to_out_ln_ofrag(const void* blended_ofrag, const void* packed_wo, const void* bias, const void* x, const void* gamma,
                        const void* beta, void* s, void* y, long R, int C, int heads, float eps, int y_qfrag, int dtype, void* stream) {
  g_sta_err[0] = 0;
  if (!blended_ofrag || !packed_wo || !x || !gamma || !beta || !s || !y) return sta_fail(STA_E_ARG, "null pointer");
  if (sta_to_out_ln_packed_wo_bytes(C, heads) == 0) return sta_fail(STA_E_UNSUP, "to_out + LayerNorm in out-fragment order: C = 320 with 8 heads only (C=%d heads=%d)", C, heads);
  if (R <= 0 || R % 16) return sta_fail(STA_E_ARG, "to_out_ln_ofrag: R=%ld (need a positive multiple of 16 rows)", R);
  if ((size_t)R * C * 2 >= 0xfffffff0ull) return sta_fail(STA_E_UNSUP, "activations must stay below 4 GiB (R=%ld)", R);
  if (dtype != STA_BF16 && dtype != STA_F16) return sta_fail(STA_E_UNSUP, "dtype %d", dtype);
  RG p{(const char*)blended_ofrag, (const char*)packed_wo, bias, gamma, beta, x, s, y, R, eps, y_qfrag ? 1 : 0};
  const long nblk = (R + 16 * RG_NW - 1) / (16 * RG_NW);
  const unsigned grid = (unsigned)(nblk < 256 ? nblk : 256);       // one persistent workgroup per CU
  hipStream_t st = (hipStream_t)stream;
  static StaLdsAttr attr_b, attr_h;
  if (dtype == STA_BF16) {
    if (!attr_b.ensure((const void*)to_out_ln_ofrag_kernel<__bf16>, RG_LDS)) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(to_out_ln) failed");
    hipLaunchKernelGGL(to_out_ln_ofrag_kernel<__bf16>, dim3(grid), dim3(64 * RG_NW), RG_LDS, st, p);
  } else {
    if (!attr_h.ensure((const void*)to_out_ln_ofrag_kernel<_Float16>, RG_LDS)) return sta_fail(STA_E_LAUNCH, "hipFuncSetAttribute(to_out_ln) failed");
    hipLaunchKernelGGL(to_out_ln_ofrag_kernel<_Float16>, dim3(grid), dim3(64 * RG_NW), RG_LDS, st, p);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? STA_OK : sta_fail(STA_E_LAUNCH, "to_out_ln_ofrag launch: %s", hipGetErrorString(e));
}

}  // extern "C"
