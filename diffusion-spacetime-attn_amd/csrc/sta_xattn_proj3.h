// sta_xattn_proj3.h — internal interface of the second-generation head-pair projection-fused forward
// (sta_xattn_proj3.hip), used by the C-ABI entry points in sta_xattn_proj.hip. Layout constants of one (ctx, head)
// block for d = 40 (SD-v1 level 0); tools/emu_pair3.py checks the same formulas on the CPU.
#ifndef STA_XATTN_PROJ3_H
#define STA_XATTN_PROJ3_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sta_p3 {
constexpr int D = 40;                       // head dim
constexpr int NT = 5;                       // 16-wide column tiles of a head pair's projection (2 * 40 = 80 columns)
constexpr int KROW = 96;                    // bytes of a K row: 4 chunks of 16 B (16x16x32 operands) + 4 units of 8 B (16x16x16)
constexpr int KR = 77;                      // K rows stored (keys 77..79 of the last key tile over-read into the V^T part)
constexpr int KBYTES = KR * KROW;           // 7392
constexpr int VROW = 160;                   // bytes of a V^T row: 2 x 64 B (keys 0..63 in S^T accumulator order) + 32 B (keys 64..79)
constexpr int VR = D + 1;                   // 40 head dims + the row of ones (softmax denominator out of the PV MFMAs)
constexpr int VBYTES = VR * VROW;           // 6560
constexpr int BLK = KBYTES + VBYTES;        // 13952 bytes per (ctx, head); a pair's two blocks are contiguous (27904)
constexpr int CTXB = 2 * BLK;
static_assert(KBYTES % 16 == 0 && BLK % 16 == 0, "16-byte LDS reads need aligned blocks");

inline bool shape_ok(int C, int heads) { return heads > 0 && heads % 2 == 0 && C == heads * D && (C == 160 || C == 320); }
__host__ __device__ constexpr int kv_region(int K) { return ((K + 2) * CTXB + 1023) / 1024 * 1024; }   // LDS bytes in front of the Wq fragments
inline int lds_bytes(int C, int K) { return kv_region(K) + NT * (C / 32) * 1024 + 16; }     // + the work-item counter
inline bool eligible(int C, int heads, int M, int K) {
  return shape_ok(C, heads) && M > 64 && M <= KR && K >= 0 && lds_bytes(C, K) <= 160 * 1024;
}
inline size_t wq_bytes(int C, int heads) { return (size_t)(heads / 2) * NT * (C / 32) * 1024; }      // pair fragments of to_q.weight
inline size_t kv_bytes(int n_ctx, int heads) { return (size_t)n_ctx * heads * BLK + 1024; }   // + slack: the last DMA piece is read whole

int pack_kv(const void* k, const void* v, void* packed, int n_ctx, int M, int C, int heads, int dtype, hipStream_t st);
int forward(const void* y, const void* wq_pair, const void* kv, const uint8_t* mask, const float* coef, void* out, int n_img,
            int N, int C, int heads, int M, int K, float sl2e, int dtype, hipStream_t st, bool qfrag = false);
}  // namespace sta_p3
#endif
