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

// Where lane row g's operand sits inside row `row` of an operand image (round 6). An MFMA operand read is one ds_read_b128 / ds_read_b64
// per lane at row pitch 96 B (K) or 160 B (V^T): 24 / 40 banks, so rows c and c + 4 of a 16-row tile started on the same banks and
// 40.6 % of the kernel's LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r05_pmc_proj_levels.txt).
// XOR-ing the slot with bits of the row index spreads the 8 (16-byte) or 16 (8-byte) lanes the LDS serves per cycle over all 32 banks:
//   16-byte chunks: slot = g ^ ((row >> 2) & 1)    8-byte units: slot = g ^ ((row >> 2) & 3)
// (an involution: the same function maps a stored slot back to the lane row, which is how pack_kv_p3_kernel uses it). Same bytes per
// block, no padding: 162816 of the CU's 163840 B stay as they were.
#ifndef STA_P3_SWIZZLE
#define STA_P3_SWIZZLE 1      // 0: the round-3 layout (same-box A/B builds: tools/lib_ab.py ... noswz=-DSTA_P3_SWIZZLE=0)
#endif
__host__ __device__ constexpr int swz_big(int g, int row) { return STA_P3_SWIZZLE ? g ^ ((row >> 2) & 1) : g; }
__host__ __device__ constexpr int swz_small(int g, int row) { return STA_P3_SWIZZLE ? g ^ ((row >> 2) & 3) : g; }

inline bool shape_ok(int C, int heads) { return heads > 0 && heads % 2 == 0 && C == heads * D && (C == 160 || C == 320); }
__host__ __device__ constexpr int kv_region(int K) { return ((K + 2) * CTXB + 1023) / 1024 * 1024; }   // LDS bytes in front of the Wq fragments
inline int lds_bytes(int C, int K) { return kv_region(K) + NT * (C / 32) * 1024 + 16; }     // + the work-item counter
inline bool eligible(int C, int heads, int M, int K) {
  return shape_ok(C, heads) && M > 64 && M <= KR && K >= 0 && lds_bytes(C, K) <= 160 * 1024;
}
inline size_t wq_bytes(int C, int heads) { return (size_t)(heads / 2) * NT * (C / 32) * 1024; }      // pair fragments of to_q.weight
inline size_t kv_bytes(int n_ctx, int heads) { return (size_t)n_ctx * heads * BLK + 1024; }   // + slack: the last DMA piece is read whole

// Head dim held by row r of a V^T image (row 40: the ones row, -1). O^T = V^T P^T leaves the MFMAs as lane (g, c) <- rows
// 16u + 4g + {0..3}: the permutation makes a lane's tile-0 and tile-1 registers 8 CONSECUTIVE dims, i.e. one 16-byte piece of
// the output row with no cross-lane exchange; head B (hp = 1) is rotated by one lane row so that the pair's 160-byte segment
// of an output row goes out as three stores: bytes 0..63 = A dims 0..31 (lane rows 0..3), bytes 64..127 = [A dims 32..39 | B
// dims 0..23] (lane row 0 | 1..3), bytes 128..159 = [B dims 24..31 | B dims 32..39] (lane rows 0 | 1) — see store_pair_rows.
__host__ __device__ constexpr int vrow_dim(int r, int hp) {
  return r == D ? -1 : (r >= 32 ? r : 8 * ((((r & 15) >> 2) + 3 * hp) & 3) + 4 * (r >> 4) + (r & 3));
}

// Out-fragment order (the kernel's OF mode; consumed by csrc/sta_rowgemm.hip): which channel of the [.., C] blended tensor
// (head-major, channel = head * 40 + dim) sits in slot j (0..7) of lane row g of fragment f (0..9) of a 16-pixel group.
//   f = 2 pr, 2 pr + 1: head 2 pr (A) / 2 pr + 1 (B): slots 0..3 = O^T rows 4g + j of tile 0, slots 4..7 = rows 16 + 4g + (j - 4)
//   f = 8 + q:          lane rows 0, 1 = pair 2q, lane rows 2, 3 = pair 2q + 1; slots 0..3 = head A's O^T row 32 + 4(g & 1) + j,
//                       slots 4..7 = head B's row 32 + 4(g & 1) + (j - 4)
__host__ __device__ constexpr int ofrag_channel(int f, int g, int j) {
  if (f < 8) {
    const int hp = f & 1, r = j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);
    return f * D + vrow_dim(r, hp);
  }
  const int pr = 2 * (f - 8) + (g >> 1), hp = j < 4 ? 0 : 1, r = 32 + 4 * (g & 1) + (j & 3);
  return (2 * pr + hp) * D + vrow_dim(r, hp);
}

int pack_kv(const void* k, const void* v, void* packed, int n_ctx, int M, int C, int heads, int dtype, hipStream_t st);
int forward(const void* y, const void* wq_pair, const void* kv, const uint8_t* mask, const float* coef, void* out, int n_img,
            int N, int C, int heads, int M, int K, float sl2e, int dtype, hipStream_t st, bool qfrag = false, bool ofrag = false,
            unsigned* stats = nullptr);
}  // namespace sta_p3
#endif
