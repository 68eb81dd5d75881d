// Micro-benchmark: how fast can every CU pull L2-resident rows into registers, by access shape?
//   mode 0  half-line gather — the B-operand shape of the projection (lane (g, c): row c, bytes 64 s + 16 g .. +15):
//           16 rows x 64 B per wave instruction
//   mode 1  full-line rows   — lane l: row l >> 3, bytes 16 (l & 7): 8 rows x 128 B per instruction
//   mode 2  linear           — 1 KiB contiguous per instruction
// Every workgroup (8 waves) walks `tiles` pixel tiles of 128 rows x 640 B (C = 320 fp16); the 8 workgroups of a
// group (same XCD after the contiguous remap) read the SAME tiles, as the 8 heads do; 2 rows (batch rows) per pixel.
// Loads are kept `DEPTH` deep in flight per wave. Prints GB/s chip-wide. Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ int xcd_remap(int bid, int nwg) { const int q = nwg / 8, xcd = bid % 8, j = bid / 8; return xcd * q + j; }

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void gather(const char* __restrict__ y, unsigned* __restrict__ sink, int rows_per_img, int tiles, int W) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int group = L >> 3;                       // 8 consecutive logical ids share their tiles
  const int img = group / W, wt = group % W;
  const char* base = y + (size_t)img * rows_per_img * 640 * 2;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 ring[DEPTH];
  // per tile and wave: 16 rows x 640 B x 2 batch rows = 20 KiB = 20 instructions of 1 KiB
  auto addr = [&](int it, int i) -> const char* {
    const int tile = wt + it * W;
    const int row0 = tile * 128 + wv * 16;
    const int r = i / 10, s = i % 10;               // batch row, 64-byte step
    const char* p = base + (size_t)r * rows_per_img * 640 + (size_t)row0 * 640;
    if (MODE == 0) return p + (size_t)c * 640 + 64 * s + 16 * g;
    if (MODE == 1) return p + (size_t)((s & 1) * 8 + (lane >> 3)) * 640 + 128 * (s >> 1) + 16 * (lane & 7);   // 5 x 128 B per row
    return p + (size_t)s * 1024 + lane * 16;
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

int main(int argc, char** argv) {
  const int imgs = 16, rows = 4096, W = 2, tiles = 16;
  const size_t bytes = (size_t)imgs * 2 * rows * 640;
  char* y; unsigned* sink;
  hipMalloc(&y, bytes); hipMalloc(&sink, 4096);
  hipMemset(y, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = imgs * W * 8;                    // 256 workgroups of 8 waves
  const double moved = (double)grid * 8 * tiles * 20 * 1024;
  auto run = [&](auto kern, const char* name) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, y, sink, rows, tiles, W);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, y, sink, rows, tiles, W);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.1f us  %7.2f TB/s into the CUs (8x re-read of %.0f MB)\n", name, ms * 1e3 / 20, moved / (ms * 1e-3 / 20) / 1e12, bytes / 1e6);
  };
  run(gather<0, 10>, "half-line gather, depth 10");
  run(gather<0, 20>, "half-line gather, depth 20");
  run(gather<1, 10>, "full-line rows, depth 10");
  run(gather<1, 20>, "full-line rows, depth 20");
  run(gather<2, 10>, "linear 1 KiB, depth 10");
  run(gather<2, 20>, "linear 1 KiB, depth 20");
  return 0;
}
