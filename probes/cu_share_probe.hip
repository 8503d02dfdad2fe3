// How many CUs does an HBM-bound streaming pass need, and can it run BESIDE an MFMA-bound kernel?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o cu_share_probe cu_share_probe.hip && ./cu_share_probe
// The training step alternates MFMA-bound kernels (implicit GEMMs, 1 block per CU, the whole register file) with
// HBM-bound BatchNorm passes (3-4 tensors of 168 MB).  They never share a CU, so overlapping them means giving each
// a SUBSET of the CUs.  This probe measures the two curves that decide whether that can pay:
//   1. stream: c[i] = f(a[i], b[i]) over 3 x 168 MB (the traffic of bn_bwd_apply) with N blocks of 1024 threads, one
//      block per CU (96 KB of dynamic LDS forces it): TB/s as a function of N;
//   2. mfma: a register-only v_mfma_f32_32x32x16_bf16 stream on M blocks of 512 threads (2 waves per SIMD, one block
//      per CU), alone and WITH the stream kernel running on the remaining CUs from a second HIP stream: does the
//      matrix pipe keep its rate (clock / power), does the stream keep its bandwidth?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// grid-stride, 16 bytes per lane, UNROLL loads of each operand in flight per thread
template <int UNROLL>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                      u32x4* __restrict__ c, size_t n) {
  extern __shared__ unsigned char lds_force[];   // occupancy control only
  const size_t stride = (size_t)gridDim.x * 1024;
  size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    u32x4 x[UNROLL], y[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { x[u] = a[i + u * stride]; y[u] = b[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      u32x4 z;
      z.x = x[u].x ^ y[u].x; z.y = x[u].y + y[u].y; z.z = x[u].z ^ y[u].w; z.w = x[u].w + y[u].z;
      c[i + u * stride] = z;
    }
  }
  for (; i < n; i += stride) {
    const u32x4 x = a[i], y = b[i];
    u32x4 z;
    z.x = x.x ^ y.x; z.y = x.y + y.y; z.z = x.z ^ y.w; z.w = x.w + y.z;
    c[i] = z;
  }
  if (n == 1) lds_force[0] = 1;
}

__global__ __launch_bounds__(512) void mfma_kernel(int iters, float* sink) {
  extern __shared__ unsigned char lds_force[];
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane + i); b[i] = (short)(0x3f00 + 2 * lane + i); }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 123.456f) { sink[0] = s; lds_force[0] = 1; }
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  const size_t bytes = 168ull << 20;             // one WRN-28-10 stage-1 activation tensor at 512 images
  const size_t n = bytes / 16;
  u32x4 *a, *b, *c;
  float* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(c, 0, bytes));
  const int LDS = 96 * 1024;
  CK(hipFuncSetAttribute((const void*)stream_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)stream_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1, f0, f1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));

  printf("# 1. streaming pass (2 x 168 MB read, 168 MB written), N blocks of 1024 threads, one per CU\n");
  const int ns[] = {16, 32, 48, 64, 96, 128, 192, 256};
  for (int unroll = 4; unroll <= 8; unroll += 4)
    for (int N : ns) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, s1));
        if (unroll == 4) stream_kernel<4><<<N, 1024, LDS, s1>>>(a, b, c, n);
        else stream_kernel<8><<<N, 1024, LDS, s1>>>(a, b, c, n);
        CK(hipEventRecord(e1, s1));
        CK(hipStreamSynchronize(s1));
        const float ms = time_ms(e0, e1);
        best = ms < best ? ms : best;
      }
      printf("stream  unroll %d  N=%3d CUs: %7.1f us  %5.2f TB/s\n", unroll, N, best * 1e3, 3.0 * bytes / best / 1e9);
    }
  {  // the usual way: many small blocks, all CUs
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, s1));
      stream_kernel<4><<<2048, 1024, 0, s1>>>(a, b, c, n);
      CK(hipEventRecord(e1, s1));
      CK(hipStreamSynchronize(s1));
      const float ms = time_ms(e0, e1);
      best = ms < best ? ms : best;
    }
    printf("stream  unroll 4  2048 blocks, no LDS (2 blocks per CU): %7.1f us  %5.2f TB/s\n", best * 1e3, 3.0 * bytes / best / 1e9);
  }

  printf("# 2. MFMA stream on M CUs (512 threads, 2 waves per SIMD), alone and beside the streaming pass on 256 - M CUs\n");
  const int iters = 1500;                        // ~ 250 us per block
  const double flop_per_block = 8.0 * iters * 16 * 32768.0;
  auto run_mfma = [&](int M, bool with_stream, int Ns, int stream_reps) {
    float best_m = 1e9f, best_s = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      mfma_kernel<<<M, 512, LDS, s1>>>(iters, sink);
      CK(hipEventRecord(e1, s1));
      if (with_stream) {
        CK(hipEventRecord(f0, s2));
        for (int k = 0; k < stream_reps; ++k) stream_kernel<8><<<Ns, 1024, LDS, s2>>>(a, b, c, n);
        CK(hipEventRecord(f1, s2));
      }
      CK(hipDeviceSynchronize());
      const float ms = time_ms(e0, e1);
      best_m = ms < best_m ? ms : best_m;
      if (with_stream) { const float t = time_ms(f0, f1) / stream_reps; best_s = t < best_s ? t : best_s; }
    }
    printf("mfma M=%3d CUs: %7.1f us  %6.0f TFLOP/s (%5.2f per CU)", M, best_m * 1e3, M * flop_per_block / best_m / 1e9,
           flop_per_block / best_m / 1e9);
    if (with_stream) printf("   | beside it, stream on %3d CUs: %7.1f us per pass  %5.2f TB/s", Ns, best_s * 1e3, 3.0 * bytes / best_s / 1e9);
    printf("\n");
  };
  run_mfma(256, false, 0, 0);
  run_mfma(224, false, 0, 0);
  run_mfma(192, false, 0, 0);
  run_mfma(128, false, 0, 0);
  run_mfma(224, true, 32, 1);
  run_mfma(192, true, 64, 2);
  run_mfma(160, true, 96, 2);
  run_mfma(128, true, 128, 3);
  run_mfma(256, true, 64, 2);      // no free CU: the stream kernel can only start when MFMA blocks leave
  return 0;
}
