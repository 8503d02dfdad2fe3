// How does the shape of a wave's fp32 atomic instruction change its cost?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_pattern_probe atomic_pattern_probe.hip && ./atomic_pattern_probe
// The weight-gradient kernels end with 255 blocks x 8 waves x 100 global_atomic_add_f32, each instruction 4 rows x 64 B
// (16 lanes = 16 consecutive cins of one (cout, tap); the 4 lane groups 4 couts apart) -- 47 MB of read-modify-write per
// launch, 30 us of a 175 us kernel.  Patterns, same addresses in total, same adds per address:
//   0  4 x 64 B   (the kernels' pattern: lane group g -> row 4 g + r, lanes -> 16 floats)
//   1  2 x 128 B  (lanes 0-31 -> 32 consecutive floats of one row, lanes 32-63 -> the next row)
//   2  1 x 256 B  (64 consecutive floats)
// 256 blocks x 512 threads, each block a private 184 KB tile (46080 floats), `splits` blocks share one tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* dw, int tiles, int tile_floats) {
  const int tile = blockIdx.x % tiles;
  float* base = dw + (size_t)tile * tile_floats;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float v = 1.0f + lane * 0.001f;
  // every wave covers tile_floats / 8 floats with 64-float instructions
  const int per_wave = tile_floats / 8;
  for (int i = 0; i < per_wave / 64; ++i) {
    int off;
    if (MODE == 0) {          // 4 rows of 16 floats, rows 288 floats (9 taps x 32 cins) apart like dw[co][tap][ci]
      const int blk = i;      // 64 floats = 4 rows x 16
      off = wave * per_wave + (blk / 2) * 128 + (blk & 1) * 16 + (lane >> 4) * 32 + (lane & 15);
    } else if (MODE == 1) {   // 2 rows of 32 floats
      off = wave * per_wave + i * 64 + lane;
    } else {
      off = wave * per_wave + i * 64 + lane;
    }
    if (MODE == 1) off = wave * per_wave + (i / 2) * 128 + (lane >> 5) * 64 + (i & 1) * 32 + (lane & 31);
    atomicAdd(base + off, v);
  }
}

int main() {
  const int tile_floats = 46080, blocks = 256;
  for (int tiles : {5, 80}) {
    float* dw;
    hipMalloc(&dw, (size_t)tiles * tile_floats * 4);
    hipMemset(dw, 0, (size_t)tiles * tile_floats * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), 0, 0, dw, tiles, tile_floats);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(512), 0, 0, dw, tiles, tile_floats);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(512), 0, 0, dw, tiles, tile_floats);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("tiles %2d (%.1f MB of gradients, %d blocks per tile)  pattern %d: %.1f us for %.1f MB of atomics\n", tiles,
             tiles * tile_floats * 4 / 1e6, blocks / tiles, mode, best * 1e3, blocks * (double)tile_floats * 4 / 1e6);
    }
    hipFree(dw);
  }
  return 0;
}
