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

// ------------------------------------------------------------------------------------------------------------
// LDS-operand MFMA streams (round 6): what do the matrix pipes sustain when every operand fragment comes
// out of LDS, for the two tilings the dense conv could use?  No LDS-DMA, no barriers, no epilogue -- only the K loop's
// ds_read_b128 + v_mfma mix, so the difference between the two is what the tiling itself buys (LDS instruction count and
// the clock the chip holds under that mix).
//   variant 0: 8 waves per CU (two per SIMD), a wave owns 64 pixels x 160 couts: per K step 4 + 10 fragment reads, 20 MFMAs
//              (conv3x3_pp_kernel's mix: 112 KB of LDS reads per CU and step);
//   variant 1: 4 waves per CU (one per SIMD), a wave owns 128 pixels x 160 couts: 8 + 10 reads, 40 MFMAs (72 KB), the
//              next step's fragments read into a second register set while this step's MFMAs issue.
// Fragment addresses follow the production swizzle (conflict-free ds_read_b128) and move through the buffers every step.
template <int MW, int NWV, int WR = 5>
__global__ __launch_bounds__(64 * NWV, NWV == 8 ? 2 : 1) void lds_mfma_kernel(int iters, float* sink) {
  constexpr int NT = 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [pixel tiles 64 KB][weight ring 3 x 10 KB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {  // fill LDS with varying bf16 (not zeros: operand toggling is what costs power)
    unsigned h = (tid + blockIdx.x * 977u) * 2654435761u + 12345u;
    for (int i = tid; i < (96 * 1024) / 4; i += 64 * NWV) {
      h = h * 1664525u + 1013904223u;
      ((unsigned*)smem)[i] = 0x3f003e80u | (h & 0x80ff80ffu);
    }
  }
  __syncthreads();
  const int frag_row = lane & 31, frag_half = lane >> 5;
  const unsigned frag = frag_row * 64 + ((frag_half ^ ((frag_row >> 2) & 3)) << 4);
  typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;
  const lds_cptr base = (lds_cptr)smem;
  f32x16 acc[NT][MW];
#pragma unroll
  for (int tn = 0; tn < NT; ++tn)
#pragma unroll
    for (int tm = 0; tm < MW; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;
  // one K HALF step (16 k): MW pixel fragments + NT weight fragments, MW * NT MFMAs
  auto load = [&](bf16x8 (&pf)[MW], bf16x8 (&wf)[NT], int half) {
    const int step = half >> 1;
    const unsigned hx = (half & 1) * 32;
    const unsigned po = ((wave * MW) * 2048 + (step & 7) * 4096) & 0xffff;      // pixel tiles: 64 KB window
    const unsigned wo = 65536 + (step % 3) * 10240;
#pragma unroll
    for (int tm = 0; tm < MW; ++tm)
      pf[tm] = *(const __attribute__((address_space(3))) bf16x8*)(base + ((po + tm * 2048 + (frag ^ hx)) & 0xffff));
#pragma unroll
    for (int tn = 0; tn < WR; ++tn)        // (WR < NT: the other fragments keep what an earlier step left in the registers)
      wf[tn] = *(const __attribute__((address_space(3))) bf16x8*)(base + wo + tn * 2048 + (frag ^ hx));
  };
  auto mfmas = [&](bf16x8 (&pf)[MW], bf16x8 (&wf)[NT]) {
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int tm = 0; tm < MW; ++tm)
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tn], pf[tm], acc[tn][tm], 0, 0, 0);
  };
  if (MW == 2) {       // two waves per SIMD: no software pipeline, the partner wave fills the gaps (production's mix)
    for (int it = 0; it < 2 * iters; it += 2) {
      bf16x8 pf0[MW], wf0[NT], pf1[MW], wf1[NT];
      if (WR < NT) {
#pragma unroll
        for (int tn = WR; tn < NT; ++tn) { wf0[tn] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8}; wf1[tn] = bf16x8{8, 7, 6, 5, 4, 3, 2, 1}; asm volatile("" : "+v"(wf0[tn]), "+v"(wf1[tn])); }
      }
      load(pf0, wf0, it);
      load(pf1, wf1, it + 1);
      mfmas(pf0, wf0);
      mfmas(pf1, wf1);
    }
  } else {             // one wave per SIMD: the next half step's fragments are read under this half step's MFMAs
    bf16x8 pfa[MW], wfa[NT], pfb[MW], wfb[NT];
    load(pfa, wfa, 0);
    for (int it = 0; it < 2 * iters; it += 2) {
      load(pfb, wfb, it + 1);
      mfmas(pfa, wfa);
#pragma unroll
      for (int i = 0; i < MW * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      load(pfa, wfa, it + 2);
      mfmas(pfb, wfb);
#pragma unroll
      for (int i = 0; i < MW * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int tn = 0; tn < NT; ++tn)
#pragma unroll
    for (int tm = 0; tm < MW; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[tn][tm][r];
  if (s == 123.456f) sink[0] = s;
}

// flops per launch = blocks x 160 MFMAs per step x iters x 32768
extern "C" int nbdt_probe_lds_mfma(int32_t blocks, int32_t iters, int32_t variant, float* sink, void* stream) {
  NBDT_REQUIRE(blocks > 0 && blocks <= 4096 && iters > 0 && iters % 2 == 0 && sink, "bad probe arguments");
  NBDT_REQUIRE(variant >= 0 && variant <= 3, "variant: 0 (8 waves x 64 pixels), 1 (4 waves x 128 pixels), 2 / 3 (variant 0 with 10 / 6 reads per step)");
  const size_t shmem = 96 * 1024;
  static nbdt::DeviceAttr site;
  if (site.need(shmem)) {
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_mfma_kernel<2, 8>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_mfma_kernel<2, 8, 3>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_mfma_kernel<2, 8, 1>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_mfma_kernel<4, 4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    site.done(shmem);
  }
  if (variant == 2) { hipLaunchKernelGGL((lds_mfma_kernel<2, 8, 3>), dim3(blocks), dim3(512), shmem, (hipStream_t)stream, iters, sink); NBDT_LAUNCH_CHECK(); return NBDT_OK; }
  if (variant == 3) { hipLaunchKernelGGL((lds_mfma_kernel<2, 8, 1>), dim3(blocks), dim3(512), shmem, (hipStream_t)stream, iters, sink); NBDT_LAUNCH_CHECK(); return NBDT_OK; }
  if (variant == 0) hipLaunchKernelGGL((lds_mfma_kernel<2, 8>), dim3(blocks), dim3(512), shmem, (hipStream_t)stream, iters, sink);
  else hipLaunchKernelGGL((lds_mfma_kernel<4, 4>), dim3(blocks), dim3(256), shmem, (hipStream_t)stream, iters, sink);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}
