// Measurement probe (gfx950 only): what the matrix pipes of THIS chip sustain, on THIS box, right now.
//
// bench.py prices every MFMA kernel against the 2.5 PFLOP/s bf16 spec.  The part is power-limited -- a kernel that keeps
// all 1024 matrix pipes busy does not run at 2.4 GHz (profiles/r02_clock_probe.txt: 1.95-2.25 PFLOP/s) -- so the bench
// line also carries two measured rates next to the spec (bench.py `roofline.attainable`, `roofline.mfma_stream`):
//   mfma_stream  this kernel: a register-only stream of independent v_mfma_f32_32x32x16_bf16, two waves per SIMD like
//                the 8-wave kernels, operands that differ per lane and per instruction (zero operands clock ~19 %
//                higher: MI355X_MICROARCH.md "DVFS give-back") -- the power / clock ceiling, no memory at all;
//   attainable   the dense kernel's own K loop in steady state: conv3x3_pp_kernel on a synthetic conv with 160
//                K chunks per tile instead of 5 (bench.py), i.e. its 20 MFMA + 14 ds_read_b128 + LDS-DMA pieces per
//                K step with exactly the real dependencies, prologue / epilogue amortised to < 4 %.
// Nothing here is on the product path; the entry point exists so that the number is measured where it is graded.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void mfma_stream_kernel(int iters, float* sink) {
  extern __shared__ unsigned char hog[];     // (sized by the launch so that one block fills a CU)
  const unsigned tid = threadIdx.x + blockIdx.x * 512u;
  bf16x8 a[4], b[4];
  unsigned h = tid * 2654435761u + 12345u;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u;
      a[k][i] = (short)(0x3f00 | ((h >> 9) & 0x80ff));       // +-[0.5, 1): bf16 with a varying mantissa and sign
      h = h * 1664525u + 1013904223u;
      b[k][i] = (short)(0x3e80 | ((h >> 9) & 0x80ff));
    }
  f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + u) & 3], b[k], acc[k], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 123.456f) sink[0] = s + hog[0];
}

// One 512-thread block per CU (`blocks` of them), each wave `iters` x 16 MFMAs = iters * 16 * 32768 flop.
extern "C" int nbdt_probe_mfma_stream(int32_t blocks, int32_t iters, float* sink, void* stream) {
  NBDT_REQUIRE(blocks > 0 && blocks <= 4096 && iters > 0 && sink, "bad probe arguments");
  const size_t shmem = 96 * 1024;            // > half of a CU's 160 KB: no second block fits beside it
  static nbdt::DeviceAttr site;
  if (site.need(shmem)) {
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_stream_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    site.done(shmem);
  }
  hipLaunchKernelGGL(mfma_stream_kernel, dim3(blocks), dim3(512), shmem, (hipStream_t)stream, iters, sink);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}
