// What does s_memtime count, and what clock does an MFMA-heavy kernel run at?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
// One 256-thread block per CU x `blocks_per_cu`; every wave runs `iters` x 16 independent
// v_mfma_f32_32x32x16_bf16 (register operands only) between two s_memtime stamps.  A SIMD retires one such MFMA per
// 32 shader cycles (16 with two waves' worth of ... no: the pipe is shared), so with one wave per SIMD
//   ticks per MFMA = 32 x (s_memtime ticks per shader cycle).
// The host times the launch with HIP events: ticks / duration = tick rate, MFMAs x 32 / duration = the shader clock if
// the pipe was saturated.  Mode 1 adds LDS reads + global loads between the MFMAs (a load closer to the conv kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(int iters, const float* src, unsigned long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
  __syncthreads();
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane + i); b[i] = (short)(0x3f00 + 2 * lane + i); }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  float extra = 0.f;
  const unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
      if (MODE == 1) {
        const float4 v = *(const float4*)(lds + ((it * 4 + u) * 64 + lane) % 1024 * 4);
        const float g = src[((size_t)blockIdx.x * 4096 + (it * 4 + u) * 64 + lane) & 0xfffff];
        extra += v.x + v.y + g;
      }
    }
  }
  const unsigned long long t1 = now();
  float s = extra;
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  int iters = 20000;
  unsigned long long* out;
  float *sink, *src;
  hipMalloc(&out, 8192 * 8);
  hipMalloc(&sink, 64);
  hipMalloc(&src, 4 << 20);
  hipMemset(src, 0, 4 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
      const int blocks = 256 * bpc;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, iters, src, out, sink);
        else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, iters, src, out, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(blocks * 4);
      hipMemcpy(h.data(), out, blocks * 4 * 8, hipMemcpyDeviceToHost);
      double ticks = 0;
      for (auto v : h) ticks += (double)v;
      ticks /= h.size();
      const double mfmas = (double)iters * 16;              // per wave
      const double flops = mfmas * 32768.0 * blocks * 4;    // 32x32x16 x 2
      printf("mode %d (%s), %d wave(s) per SIMD: %.2f ms, %.0f TFLOP/s, %.1f ticks per MFMA per wave, %.3f ticks/ns;"
             "  if the pipe is full the shader clock is %.3f GHz (MFMAs per SIMD x 32 cycles / time)\n",
             mode, mode ? "MFMA + LDS read + global load" : "MFMA only", bpc, ms, flops / ms / 1e9, ticks / mfmas,
             ticks / (ms * 1e6), mfmas * bpc * 32.0 / (ms * 1e6));
    }
  return 0;
}
