// sta_selfattn_dev.h — device-side pieces shared by the self-attention kernels of libsta_xattn.so
// (sta_selfattn.hip: forward; sta_selfattn_bwd.hip: backward). gfx950 only; see sta_selfattn.hip for the operand layout.
#ifndef STA_SELFATTN_DEV_H
#define STA_SELFATTN_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sta_xattn.h"
#include "sta_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename T> struct Tr;
template <> struct Tr<__bf16> {
  using V8 = bf16x8;
  using V4 = bf16x4;
  static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Tr<_Float16> {
  using V8 = f16x8;
  using V4 = f16x4;
  static __device__ __forceinline__ f32x4 mfma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int KB = 64;        // keys per block
constexpr float RESCALE_LOG2 = 8.0f;   // see the deferred rescale in the kernel
constexpr int FRAG = 1024;    // bytes per fragment

// key (within a 64-key block) held by row i of S^T tile T
__device__ __forceinline__ int tile_key(int T, int i) { return 32 * (T >> 1) + 8 * (i >> 2) + 4 * (T & 1) + (i & 3); }

}  // namespace
#endif
