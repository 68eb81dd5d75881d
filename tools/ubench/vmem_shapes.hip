// Micro-benchmark: what one vector-memory instruction costs a CU by the way its 64 lanes are laid over the rows.
// Round 3 question: the level-0 kernel is bound by its load + store skeleton (profiles/r03_level0.md) — is it the BYTES, the
// LINES, or the way the lanes of one row are spread over the wave? Same bytes in every mode:
//   loads  (L2-resident: 4 workgroups of a group read the same rows, as the 4 head pairs do)
//     L0  MFMA B-operand shape: lane (g, c) -> row c, bytes 16 g .. of a 64-byte step          (16 rows x 64 B, lanes of a row 16 apart)
//     L1  the same 16 rows x 64 B with the 4 lanes of a row ADJACENT: lane l -> row l >> 2, bytes 16 (l & 3)
//     L2  8 rows x 128 B, lanes of a row adjacent: lane l -> row l >> 3, bytes 16 (l & 7)
//     L3  8 rows x 128 B in the DPP-friendly shape of the kernel: lane (g, c) -> row c & 7, slot g + 4 (c >> 3)
//   stores (each workgroup writes its own 160-byte pair segment of every row, as the pair kernel does)
//     S0  lane (g, c) -> row c, bytes 16 g of a 64-byte piece                                   (lanes of a row 16 apart)
//     S1  lane l -> row l >> 2, bytes 16 (l & 3)                                                (adjacent)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/vmem_shapes.hip -o build/vmem_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ int xcd_remap(int bid, int nwg) { const int q = nwg / 8, xcd = bid % 8, j = bid / 8; return xcd * q + j; }

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void loads(const char* __restrict__ y, unsigned* __restrict__ sink, int rows_per_img, int tiles, int W) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int group = L >> 2;                       // 4 consecutive logical ids (the 4 head pairs) share their tiles
  const int img = group / W, wt = group % W;
  const char* base = y + (size_t)img * rows_per_img * 640 * 2;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 ring[DEPTH];
  auto addr = [&](int it, int i) -> const char* {   // i = 0..19: the 20 instructions of a (tile, wave): 16 rows x 640 B x 2 batch rows
    const int tile = wt + it * W;
    const int row0 = tile * 128 + wv * 16;
    const int r = i / 10, s = i % 10;
    const char* p = base + (size_t)r * rows_per_img * 640 + (size_t)row0 * 640;
    if (MODE == 0) return p + (size_t)c * 640 + 64 * s + 16 * g;
    if (MODE == 1) return p + (size_t)(lane >> 2) * 640 + 64 * s + 16 * (lane & 3);
    if (MODE == 2) return p + (size_t)((s & 1) * 8 + (lane >> 3)) * 640 + 128 * (s >> 1) + 16 * (lane & 7);
    return p + (size_t)((s & 1) * 8 + (c & 7)) * 640 + 128 * (s >> 1) + 16 * (g + 4 * (c >> 3));
  };
  const int total = tiles * 20;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) ring[d] = *(const u32x4*)addr(d / 20, d % 20);
  for (int n = 0; n < total; n += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc += ring[d];
      const int m = n + d + DEPTH;
      if (m < total) ring[d] = *(const u32x4*)addr(m / 20, m % 20);
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) sink[blockIdx.x] = acc[0];
}

template <int MODE>
__global__ __launch_bounds__(512) void stores(char* __restrict__ out, int rows_per_img, int tiles, int W) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int group = L >> 2, pr = L & 3;
  const int img = group / W, wt = group % W;
  char* base = out + (size_t)img * rows_per_img * 640 * 2 + 160 * pr;
  const u32x4 v = {(unsigned)lane, (unsigned)wv, 3u, 4u};
  for (int it = 0; it < tiles; ++it) {
    const int row0 = (wt + it * W) * 128 + wv * 16;
    for (int r = 0; r < 2; ++r) {
      char* p = base + (size_t)r * rows_per_img * 640 + (size_t)row0 * 640;
      // 160 bytes per row = pieces of 64 + 64 + 32 bytes: three instructions (the last one half empty), as the kernel's
      for (int q = 0; q < 3; ++q) {
        const int row = MODE == 0 ? c : (lane >> 2), sl = MODE == 0 ? g : (lane & 3);
        if (q < 2 || sl < 2) *(u32x4*)(p + (size_t)row * 640 + 64 * q + 16 * sl) = v;
      }
    }
  }
}

// Mixed skeletons: every (tile, wave) loads its 20 KiB of y (one item ahead, like the kernel's ring) and stores its output.
//   MIX 0  pair design, MFMA-shape loads: 4 workgroups read the same tile, each stores its 160-byte segment of every row (6 stores)
//   MIX 1  pair design, adjacent-lane loads
//   MIX 2  all-heads design: ONE workgroup per tile reads y once and stores whole 640-byte rows (20 stores of 16 rows x 64 B);
//          `extra` linear 1-KiB loads per (tile, wave) stand for the operand images it would have to stream (L2-resident, 632 KiB
//          per tile and workgroup = 79 per wave)
template <int MIX>
__global__ __launch_bounds__(512) void mixed(const char* __restrict__ y, char* __restrict__ out, const char* __restrict__ ops, unsigned* __restrict__ sink,
                                             int rows_per_img, int tiles, int W, int extra) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int group = MIX == 2 ? L : (L >> 2), pr = MIX == 2 ? 0 : (L & 3);
  const int img = group / W, wt = group % W;
  const char* base = y + (size_t)img * rows_per_img * 640 * 2;
  char* obase = out + (size_t)img * rows_per_img * 640 * 2 + 160 * pr;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 ring[20];
  auto addr = [&](int it, int i) -> const char* {
    const int row0 = (wt + it * W) * 128 + wv * 16;
    const int r = i / 10, s = i % 10;
    const char* p = base + (size_t)r * rows_per_img * 640 + (size_t)row0 * 640;
    if (MIX == 0) return p + (size_t)c * 640 + 64 * s + 16 * g;
    return p + (size_t)(lane >> 2) * 640 + 64 * s + 16 * (lane & 3);
  };
#pragma unroll
  for (int d = 0; d < 20; ++d) ring[d] = *(const u32x4*)addr(0, d);
  for (int it = 0; it < tiles; ++it) {
#pragma unroll
    for (int d = 0; d < 20; ++d) {
      acc += ring[d];
      if (it + 1 < tiles) ring[d] = *(const u32x4*)addr(it + 1, d);
    }
    for (int e = 0; e < extra; ++e) acc += *(const u32x4*)(ops + ((size_t)((e * 8 + wv) % 632) * 1024) + lane * 16);
    const int row0 = (wt + it * W) * 128 + wv * 16;
    for (int r = 0; r < 2; ++r) {
      char* p = obase + (size_t)r * rows_per_img * 640 + (size_t)row0 * 640;
      if (MIX == 2) {
        for (int q = 0; q < 10; ++q) *(u32x4*)(p + (size_t)(lane >> 2) * 640 + 64 * q + 16 * (lane & 3)) = acc;
      } else {
        for (int q = 0; q < 3; ++q)
          if (q < 2 || g < 2) *(u32x4*)(p + (size_t)c * 640 + 64 * q + 16 * g) = acc;
      }
    }
  }
  if (acc[0] == 0x12345678u) sink[blockIdx.x] = acc[0];
}

int main() {
  const int imgs = 32, rows = 4096, W = 2, tiles = 16;          // 32 images x 2 tile groups x 4 pairs = 256 workgroups x 16 tiles
  const size_t bytes = (size_t)imgs * 2 * rows * 640;
  char *y, *o; unsigned* sink;
  hipMalloc(&y, bytes); hipMalloc(&o, bytes); hipMalloc(&sink, 4096);
  hipMemset(y, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = imgs * W * 4;
  auto timeit = [&](auto launch, const char* name, double moved) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.1f us  %6.2f TB/s\n", name, ms * 1e3 / 20, moved / (ms * 1e-3 / 20) / 1e12);
  };
  const double ld = (double)grid * 8 * tiles * 20 * 1024, st = (double)bytes;
  timeit([&] { hipLaunchKernelGGL((loads<0, 10>), dim3(grid), dim3(512), 0, 0, y, sink, rows, tiles, W); }, "L0 MFMA shape (16 rows x 64 B, spread)", ld);
  timeit([&] { hipLaunchKernelGGL((loads<1, 10>), dim3(grid), dim3(512), 0, 0, y, sink, rows, tiles, W); }, "L1 16 rows x 64 B, adjacent lanes", ld);
  timeit([&] { hipLaunchKernelGGL((loads<2, 10>), dim3(grid), dim3(512), 0, 0, y, sink, rows, tiles, W); }, "L2 8 rows x 128 B, adjacent lanes", ld);
  timeit([&] { hipLaunchKernelGGL((loads<3, 10>), dim3(grid), dim3(512), 0, 0, y, sink, rows, tiles, W); }, "L3 8 rows x 128 B, DPP shape (spread)", ld);
  timeit([&] { hipLaunchKernelGGL((loads<2, 20>), dim3(grid), dim3(512), 0, 0, y, sink, rows, tiles, W); }, "L2 depth 20", ld);
  timeit([&] { hipLaunchKernelGGL((stores<0>), dim3(grid), dim3(512), 0, 0, o, rows, tiles, W); }, "S0 stores, lanes of a row 16 apart", st);
  timeit([&] { hipLaunchKernelGGL((stores<1>), dim3(grid), dim3(512), 0, 0, o, rows, tiles, W); }, "S1 stores, adjacent lanes", st);
  char* opsb; hipMalloc(&opsb, 1 << 20); hipMemset(opsb, 1, 1 << 20);
  const double both = ld + st;
  timeit([&] { hipLaunchKernelGGL((mixed<0>), dim3(grid), dim3(512), 0, 0, y, o, opsb, sink, rows, tiles, W, 0); }, "MIX0 pair design, MFMA-shape loads + 6 stores", both);
  timeit([&] { hipLaunchKernelGGL((mixed<1>), dim3(grid), dim3(512), 0, 0, y, o, opsb, sink, rows, tiles, W, 0); }, "MIX1 pair design, adjacent loads + 6 stores", both);
  // all heads: 256 workgroups, each owns its tiles alone: 32 images x 32 tiles / 256 = 4 tiles per workgroup (W = 8 groups per image)
  const double once = (double)bytes * 2;
  timeit([&] { hipLaunchKernelGGL((mixed<2>), dim3(256), dim3(512), 0, 0, y, o, opsb, sink, rows, 4, 8, 0); }, "MIX2 all heads: y once + full-row stores", once);
  timeit([&] { hipLaunchKernelGGL((mixed<2>), dim3(256), dim3(512), 0, 0, y, o, opsb, sink, rows, 4, 8, 79); }, "MIX2 + 79 operand loads per (tile, wave)", once + 256.0 * 8 * 4 * 79 * 1024);
  return 0;
}
