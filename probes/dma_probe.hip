// LDS-DMA (global_load_lds_dwordx4) throughput / issue-cost probe for gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip && ./dma_probe
// One 512-thread block per CU.  `issuers` waves stream `iters` bursts of `burst` 1-KiB pieces each from an
// L2-resident source (src_bytes per CU-group, re-read by every block) into LDS, waiting vmcnt(0) after each burst.
// shape 0: a piece is one contiguous KiB; shape 1: 16 rows x 64 B at a 2880-B stride (a 32-channel slice of 16
// pixels of a 160-channel NHWC tensor: what the conv kernels' halo / untiled weight pieces look like).
// mode 0: LDS-DMA; mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int MODE, int SHAPE, int BURST>
__global__ __launch_bounds__(512, 2) void probe(const char* src, unsigned src_bytes, int issuers, int iters,
                                                unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // shape 2: 64 pixels x 16 B at a 320-B pixel stride (the weight-gradient kernel's x pieces: 8 channels per pixel);
  // shape 3: 32 pixels x 2 x 16 B (its gy pieces: lanes 0-31 one 8-channel chunk, lanes 32-63 the next chunk)
  const unsigned voff = SHAPE == 0 ? lane * 16u
                      : SHAPE == 1 ? (unsigned)((lane >> 2) * 2880 + (lane & 3) * 16)
                      : SHAPE == 2 ? (unsigned)(lane * 320)
                                   : (unsigned)((lane & 31) * 320 + (lane >> 5) * 16);
  const unsigned piece_bytes = SHAPE == 0 ? 1024u : SHAPE == 1 ? 16u * 2880u : SHAPE == 2 ? 64u * 320u : 32u * 320u;
  const unsigned npieces = src_bytes / piece_bytes;
  unsigned long long t_issue = 0, t_total = 0;
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  u4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = now();
  if (wave < issuers) {
    unsigned p = (blockIdx.x * 7 + wave * 131) % npieces;
    for (int it = 0; it < iters; ++it) {
      const unsigned long long a = now();
      if (MODE == 0) {
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
          const char* sb = src + (size_t)__builtin_amdgcn_readfirstlane(p) * piece_bytes;
          glds16_s(sb, voff, lds_base + ((wave * BURST + b) & 63) * 1024);
          p = (p + issuers * 17) % npieces;
        }
        const unsigned long long c = now();
        t_issue += c - a;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        u4 v[BURST];
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
          v[b] = *(const u4*)(src + (size_t)__builtin_amdgcn_readfirstlane(p) * piece_bytes + voff);
          p = (p + issuers * 17) % npieces;
        }
        const unsigned long long c = now();
        t_issue += c - a;
#pragma unroll
        for (int b = 0; b < BURST; ++b) *(u4*)(smem + ((wave * BURST + b) & 63) * 1024 + lane * 16) = v[b];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
    }
  }
  __syncthreads();
  t_total = now() - t0;
  if (lane == 0) {
    out[(blockIdx.x * 8 + wave) * 2] = t_issue;
    out[(blockIdx.x * 8 + wave) * 2 + 1] = t_total;
  }
  if (sink && acc[0] == 123) sink[0] = smem[lane];
}

template <int MODE, int SHAPE, int BURST>
static void run(const char* src, unsigned src_bytes, int issuers, int iters, unsigned long long* d_out, const char* tag) {
  const int blocks = 256;
  hipFuncSetAttribute((const void*)probe<MODE, SHAPE, BURST>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, SHAPE, BURST><<<blocks, 512, 65536>>>(src, src_bytes, issuers, iters, d_out, nullptr);
  hipEventRecord(e0);
  probe<MODE, SHAPE, BURST><<<blocks, 512, 65536>>>(src, src_bytes, issuers, iters, d_out, nullptr);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 8 * 2);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  double issue = 0, total = 0;
  int n = 0;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < issuers; ++w) { issue += h[(b * 8 + w) * 2]; total += h[(b * 8 + w) * 2 + 1]; ++n; }
  issue /= n; total /= n;
  const double pieces_per_wave = (double)iters * BURST;
  const double bytes_cu = pieces_per_wave * issuers * 1024.0;
  printf("%-28s issuers %d burst %2d: issue %6.1f cyc/piece   %6.2f B/clk/CU (in-kernel)   %7.1f GB/s/CU  chip %6.2f TB/s  (%.1f us)\n",
         tag, issuers, BURST, issue / pieces_per_wave, bytes_cu / total, bytes_cu / (ms * 1e-3) / 1e9,
         bytes_cu * 256 / (ms * 1e-3) / 1e12, ms * 1e3);
}

int main() {
  const unsigned src_bytes = 16 * 2880 * 64;   // 2.9 MB: L2-resident, shared by every block (like a layer's weights)
  char* src;
  hipMalloc(&src, src_bytes + 65536);
  hipMemset(src, 1, src_bytes + 65536);
  unsigned long long* d_out;
  hipMalloc(&d_out, 256 * 8 * 2 * 8);
  const int iters = 200;
  for (int issuers : {2, 4, 8}) {
    run<0, 0, 1>(src, src_bytes, issuers, iters * 4, d_out, "LDS-DMA contiguous KiB");
    run<0, 0, 4>(src, src_bytes, issuers, iters, d_out, "LDS-DMA contiguous KiB");
    run<0, 0, 16>(src, src_bytes, issuers, iters / 4, d_out, "LDS-DMA contiguous KiB");
    run<0, 1, 4>(src, src_bytes, issuers, iters, d_out, "LDS-DMA 16 x 64 B strided");
    run<0, 1, 16>(src, src_bytes, issuers, iters / 4, d_out, "LDS-DMA 16 x 64 B strided");
    run<0, 2, 4>(src, src_bytes, issuers, iters, d_out, "LDS-DMA 64 x 16 B strided");
    run<0, 2, 16>(src, src_bytes, issuers, iters / 4, d_out, "LDS-DMA 64 x 16 B strided");
    run<0, 3, 4>(src, src_bytes, issuers, iters, d_out, "LDS-DMA 32 x 2 x 16 B strided");
    run<0, 3, 16>(src, src_bytes, issuers, iters / 4, d_out, "LDS-DMA 32 x 2 x 16 B strided");
    run<1, 0, 4>(src, src_bytes, issuers, iters, d_out, "global_load -> ds_write KiB");
    run<1, 1, 4>(src, src_bytes, issuers, iters, d_out, "global_load -> ds_write 16x64B");
  }
  return 0;
}
