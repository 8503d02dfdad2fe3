// Does hipExtStreamCreateWithCUMask give a HARD CU partition on MI355X, and do two masked streams overlap?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o cu_mask_probe cu_mask_probe.hip && ./cu_mask_probe
// The engine's CU sharing (HBM-bound BatchNorm passes beside MFMA-bound weight gradients) confines kernels by grid
// size + a 96 KB LDS request + the dispatcher's round-robin of blocks over the 8 XCDs.  A CU-masked stream would be the
// API made for this.  The probe answers, by reading HW_ID / XCC_ID from every block:
//   1. which physical CUs (xcc, se, cu) a kernel lands on for a given mask (bit i of the mask -> which CU?);
//   2. whether a masked stream's kernels stay inside the mask when the grid is larger than the mask;
//   3. whether a streaming kernel on one masked stream and an MFMA kernel on the complementary one run at the same
//      time and at the rates the grid-size confinement measured (probes/cu_share_probe.hip);
//   4. what a third party holding a few CUs (an RCCL all-reduce kernel) does to a one-block-per-CU persistent kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void where_kernel(unsigned* out, int spin) {
  extern __shared__ unsigned char lds_force[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) { }      // 100 MHz ticks
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  if (spin < 0) lds_force[0] = 1;
}

template <int UNROLL>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                      u32x4* __restrict__ c, size_t n) {
  extern __shared__ unsigned char lds_force[];
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

// a third party that holds `blocks` CU slots for `ticks` of the 100 MHz wall clock with few registers and no LDS
// (an RCCL kernel: a few hundred threads per channel) -- a whole-register-file block cannot join it on its CU
__global__ __launch_bounds__(256) void hold_kernel(long long ticks) {
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)ticks) { }
}

static float ms_of(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

struct Where { std::set<unsigned> cus; int per_xcc[8] = {0}; };
static Where decode(const std::vector<unsigned>& v, int blocks) {
  Where w;
  for (int i = 0; i < blocks; ++i) {
    const unsigned hw = v[2 * i], xcc = v[2 * i + 1] & 0xf;
    const unsigned key = (xcc << 16) | (hw & 0xff00);     // se_id[15:13] sh_id[12] cu_id[11:8]
    if (w.cus.insert(key).second && xcc < 8) w.per_xcc[xcc]++;
  }
  return w;
}

int main() {
  const int LDS = 96 * 1024;
  CK(hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)stream_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  unsigned* d_out;
  CK(hipMalloc(&d_out, 8192 * 8));
  std::vector<unsigned> h(8192 * 2);
  auto where = [&](hipStream_t s, int blocks, const char* tag) {
    CK(hipMemsetAsync(d_out, 0xff, 8192 * 8, s));
    where_kernel<<<blocks, 256, LDS, s>>>(d_out, 2000);     // 20 us: every block of one round is resident at once
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_out, 8192 * 8, hipMemcpyDeviceToHost));
    Where w = decode(h, blocks);
    printf("%-58s %4d blocks -> %3zu distinct CUs; per XCC:", tag, blocks, w.cus.size());
    for (int x = 0; x < 8; ++x) printf(" %2d", w.per_xcc[x]);
    printf("\n");
    return w;
  };
  hipStream_t plain;
  CK(hipStreamCreate(&plain));
  printf("# 1. where do blocks land (one 96-KB-LDS block per CU, 20 us each)\n");
  where(plain, 256, "plain stream");
  where(plain, 64, "plain stream");
  where(plain, 96, "plain stream");

  auto masked = [&](const std::vector<unsigned>& m, hipStream_t* s) {
    hipError_t e = hipExtStreamCreateWithCUMask(s, (unsigned)m.size(), m.data());
    if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e)); return false; }
    return true;
  };
  auto bits = [](std::vector<unsigned>& m, int lo, int hi, int step = 1) {
    for (int i = lo; i < hi; i += step) m[i >> 5] |= 1u << (i & 31);
  };
  printf("# 2. CU-masked streams (mask = 8 x 32 bits)\n");
  {
    std::vector<unsigned> m(8, 0);
    bits(m, 0, 64);
    hipStream_t s;
    if (masked(m, &s)) {
      where(s, 64, "mask bits 0..63");
      where(s, 256, "mask bits 0..63, grid of 256 (4 rounds if the mask holds)");
      CK(hipStreamDestroy(s));
    }
  }
  {
    std::vector<unsigned> m(8, 0);
    bits(m, 0, 256, 4);
    hipStream_t s;
    if (masked(m, &s)) { where(s, 64, "mask every 4th bit (64 CUs)"); where(s, 256, "mask every 4th bit, grid of 256"); CK(hipStreamDestroy(s)); }
  }
  {
    std::vector<unsigned> m(8, 0);
    bits(m, 0, 8);
    hipStream_t s;
    if (masked(m, &s)) { where(s, 8, "mask bits 0..7"); CK(hipStreamDestroy(s)); }
  }
  {
    std::vector<unsigned> m(8, 0);
    bits(m, 0, 32);
    hipStream_t s;
    if (masked(m, &s)) { where(s, 32, "mask bits 0..31"); CK(hipStreamDestroy(s)); }
  }

  // ---- 3. overlap on complementary masks
  const size_t bytes = 168ull << 20, n = bytes / 16;
  u32x4 *a, *b, *c;
  float* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(c, 0, bytes));
  hipEvent_t e0, e1, f0, f1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  const int iters = 1500;
  const double flop_per_block = 8.0 * iters * 16 * 32768.0;
  auto pair = [&](hipStream_t sm, hipStream_t ss, int M, int Ns, int reps, const char* tag) {
    float best_m = 1e9f, best_s = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, sm));
      mfma_kernel<<<M, 512, LDS, sm>>>(iters, sink);
      CK(hipEventRecord(e1, sm));
      if (Ns > 0) {
        CK(hipEventRecord(f0, ss));
        for (int k = 0; k < reps; ++k) stream_kernel<8><<<Ns, 1024, LDS, ss>>>(a, b, c, n);
        CK(hipEventRecord(f1, ss));
      }
      CK(hipDeviceSynchronize());
      best_m = fminf(best_m, ms_of(e0, e1));
      if (Ns > 0) best_s = fminf(best_s, ms_of(f0, f1) / reps);
    }
    printf("%-44s mfma %3d blocks: %7.1f us %6.0f TFLOP/s", tag, M, best_m * 1e3, M * flop_per_block / best_m / 1e9);
    if (Ns > 0) printf(" | stream %3d blocks: %7.1f us per pass %5.2f TB/s", Ns, best_s * 1e3, 3.0 * bytes / best_s / 1e9);
    printf("\n");
  };
  printf("# 3. MFMA stream + streaming pass at the same time: grid-size confinement vs complementary CU masks\n");
  hipStream_t p2;
  CK(hipStreamCreate(&p2));
  pair(plain, p2, 192, 64, 2, "plain streams, grids 192 + 64");
  pair(plain, p2, 160, 96, 2, "plain streams, grids 160 + 96");
  for (int ns : {64, 96}) {
    // complementary masks with ns/8 CUs per XCD for the stream, assuming bit i -> XCD i % 8 (checked in part 2)
    std::vector<unsigned> ms(8, 0), mm(8, 0);
    for (int i = 0; i < 256; ++i) {
      const bool to_stream = (i / 8) < ns / 8;      // the first ns/8 "rows" of 8 consecutive bits
      (to_stream ? ms : mm)[i >> 5] |= 1u << (i & 31);
    }
    hipStream_t s_s, s_m;
    if (masked(ms, &s_s) && masked(mm, &s_m)) {
      char tag[96];
      snprintf(tag, sizeof(tag), "masked streams %d + %d CUs, grids %d + %d", 256 - ns, ns, 256 - ns, ns);
      where(s_s, ns, "  (stream-side mask)");
      where(s_m, 256 - ns, "  (mfma-side mask)");
      pair(s_m, s_s, 256 - ns, ns, 2, tag);
      snprintf(tag, sizeof(tag), "masked streams %d + %d CUs, grids 256 + %d", 256 - ns, ns, ns);
      pair(s_m, s_s, 256, ns, 2, tag);          // a full-chip grid on the masked stream: rounds instead of stealing
      CK(hipStreamDestroy(s_s)); CK(hipStreamDestroy(s_m));
    }
  }

  // ---- 4. a third party holding CUs while a one-block-per-CU kernel runs
  printf("# 4. one-block-per-CU MFMA kernel (256 blocks, ~800 us) with a holder of k blocks x 256 threads for 400 us on another stream\n");
  hipStream_t p3;
  CK(hipStreamCreate(&p3));
  for (int k : {0, 8, 16, 32, 64}) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      if (k > 0) hold_kernel<<<k, 256, 0, p3>>>(40000);
      CK(hipEventRecord(e0, plain));
      mfma_kernel<<<256, 512, LDS, plain>>>(iters, sink);
      CK(hipEventRecord(e1, plain));
      CK(hipDeviceSynchronize());
      best = fminf(best, ms_of(e0, e1));
    }
    printf("holder k=%2d: mfma kernel %7.1f us\n", k, best * 1e3);
  }
  for (int k : {8, 32}) {      // the same with the MFMA kernel sized for the CUs the holder leaves
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      hold_kernel<<<k, 256, 0, p3>>>(40000);
      CK(hipEventRecord(e0, plain));
      mfma_kernel<<<256 - k, 512, LDS, plain>>>(iters * 256 / (256 - k), sink);
      CK(hipEventRecord(e1, plain));
      CK(hipDeviceSynchronize());
      best = fminf(best, ms_of(e0, e1));
    }
    printf("holder k=%2d, mfma grid %3d with the same total work: %7.1f us\n", k, 256 - k, best * 1e3);
  }
  return 0;
}
