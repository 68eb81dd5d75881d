// sta_xattn_proj2.h — internal interface of the head-pair projection-fused forward (sta_xattn_proj2.hip), used by the
// C-ABI entry points in sta_xattn_proj.hip. Constants of the compact (ctx, head) block for d = 40.
#ifndef STA_XATTN_PROJ2_H
#define STA_XATTN_PROJ2_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace sta_pair {
constexpr int D = 40;                       // head dim this kernel is built for (SD-v1 level 0)
constexpr int NT = 5;                       // 16-wide column tiles of a head pair's projection (2 * 40 = 80 columns)
constexpr int KROWS = 80;                   // key rows of a K block (77 real, the rest zero)
constexpr int KBYTES = KROWS * D * 2;       // 6400
constexpr int VROWS = D + 1;                // 40 head dims + the row of ones
constexpr int VSLOTS = 88;                  // key slots per V^T row (>= 80): row stride 176 B = 44 dwords = 4 x odd, so the
constexpr int VS = VSLOTS * 2;              //   16 rows a ds_read_b64 touches start in 16 different 4-bank groups (no
constexpr int VBYTES = VROWS * VS;          //   conflicts; 168 B measured 26 % conflict cycles). 7216 bytes
constexpr int BLK = 13824;                  // one (ctx, head) block, K then V^T; a context's two blocks are 27 KiB
constexpr int SLACK = 2048;                 // zeroed LDS behind the last block (over-reads of the unused V^T rows 41..47)
static_assert(KBYTES + VBYTES <= BLK && (2 * BLK) % 1024 == 0, "block layout");

inline bool shape_ok(int C, int heads) { return heads > 0 && heads % 2 == 0 && C == heads * D && C % 160 == 0; }
inline int lds_bytes(int C, int K) { return NT * (C / 32) * 1024 + (K + 2) * 2 * BLK + SLACK; }
inline bool eligible(int C, int heads, int M, int K) {
  return shape_ok(C, heads) && M > 64 && M <= 80 && K >= 0 && lds_bytes(C, K) <= 160 * 1024;
}
inline size_t wq_bytes(int C, int heads) { return (size_t)(heads / 2) * NT * (C / 32) * 1024; }
inline size_t kv_bytes(int n_ctx, int heads) { return (size_t)n_ctx * heads * BLK; }

int pack_kv(const void* k, const void* v, void* packed, int n_ctx, int M, int C, int heads, int dtype, hipStream_t st);
int forward(const void* y, const void* wq_pair, const void* kv_pair, const uint8_t* mask, const float* coef, void* out, int n_img,
            int N, int C, int heads, int M, int K, float sl2e, int dtype, hipStream_t st);
}  // namespace sta_pair
#endif
