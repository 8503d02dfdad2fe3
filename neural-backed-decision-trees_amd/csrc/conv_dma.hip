// Implicit-GEMM convolution v2: LDS-DMA staging with a 3-stage ring (gfx950 only).
//
// Same contract, tiling, MFMA mapping, LDS swizzle and epilogue as conv_igemm_kernel (conv.hip);
// what changes is how tiles reach LDS:
//
//   v1: global_load_dwordx4 -> 28 staging VGPRs -> ds_write_b128, 2 LDS buffers, 1-deep prefetch,
//       hipcc-placed waits.  rocprofv3: 33% SQ_WAIT_ANY + 41% SQ_WAIT_INST_ANY, 33% of MFMA peak.
//   v2: global_load_lds_dwordx4 (no staging registers, no ds_write), 3 LDS stages, the DMA for K tile
//       t+2 is issued right after the barrier that frees its slot, counted s_waitcnt vmcnt(N) + one raw
//       s_barrier per K tile (the protocol proven in wgrad_dma.hip).
//
// global_load_lds writes LDS at (wave-uniform base + lane*16), so the [row][4 x 16 B] tile image is
// linear; the bank-conflict swizzle moves to the SOURCE side: LDS position (row, c) receives data
// chunk c ^ ((row>>2)&3), and fragment reads look data chunk c up at position c ^ ((row>>2)&3)
// (same involution on both sides, cdna guide rule 21).
// One wave-instruction = 16 rows x 64 B.  Each lane's pixel/weight-row offsets are fixed for the whole
// block, so a DMA costs one 64-bit add.  The per-tap input offset lives in lane `tap` of a VGPR and is
// fetched with v_readlane (no scalar memory load inside the pipeline).
#include "conv_common.h"

// fewest DMAs any wave issues per stage: A = 16 instructions (4 per wave), W = BN/16 instructions
// handed out as {(w+2)%4, +4, ..}
constexpr int min_w_dma(int w_instr) {
  int best = 1 << 30;
  for (int w = 0; w < 4; ++w) {
    int n = 0;
    for (int id = (w + 2) & 3; id < w_instr; id += 4) ++n;
    best = n < best ? n : best;
  }
  return best;
}

template <int NT, bool HAS_RES, int STATS>
__device__ __forceinline__ void conv_igemm_dma_body(const nbdt::ConvDmaParams& p, const int bid) {
  constexpr int BN = 32 * NT;
  constexpr int A_BYTES = BM * BK * 2;  // 16 KiB
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + W_BYTES;
  constexpr int A_INSTR = A_BYTES / 1024, W_INSTR = W_BYTES / 1024;
  constexpr int IPW_W = (W_INSTR + 3) / 4;
  constexpr int MINPW = A_INSTR / 4 + min_w_dma(W_INSTR);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int item = (bid & 7) * p.per_xcd + (bid >> 3);
  if (item >= p.m_blocks * p.n_blocks) return;
  const int m_blk = item / p.n_blocks;
  const int n_blk = item - m_blk * p.n_blocks;
  const int m0 = m_blk * BM;
  const int n0 = n_blk * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const nbdt_conv_desc& d = p.d;

#define NBDT_PIN(x) __builtin_amdgcn_readfirstlane(x)
  const int cin = NBDT_PIN(d.cin), ntaps = NBDT_PIN(d.ntaps);
  const int kchunks = cin >> 5;
  const int nk = ntaps * kchunks;
  const int w_row_len = d.w_ntaps * cin;
  const unsigned long long in_u = (unsigned long long)p.in, w_u = (unsigned long long)p.w;
  const bf16_t* in_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(in_u >> 32)) << 32) |
                                          (unsigned)NBDT_PIN((unsigned)in_u));
  const bf16_t* w_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(w_u >> 32)) << 32) |
                                         (unsigned)NBDT_PIN((unsigned)w_u));
#undef NBDT_PIN
  // lane t (< ntaps) keeps tap t's input offset and weight k-offset; fetched with v_readlane
  const int tap_in_v = lane < ntaps ? d.tap_off[lane < 9 ? lane : 0] : 0;
  const int tap_w_v = lane < ntaps ? d.w_tap[lane < 9 ? lane : 0] * cin : 0;

  // ---- DMA slot tables (fixed per block): position (row, cpos) receives data chunk cpos ^ swz(row)
  const int cpos = lane & 3;
  int a_src[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = (wave + 4 * k) * 16 + (lane >> 2);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    a_src[k] = pix_offset(m, d.gh, d.gw, d.in_bs, d.in_hs, d.in_ws, d.in_base) + ((cpos ^ ((row >> 2) & 3)) << 3);
  }
  int w_src[IPW_W];
#pragma unroll
  for (int k = 0; k < IPW_W; ++k) {
    const int id = ((wave + 2) & 3) + 4 * k;
    int row = id * 16 + (lane >> 2);
    row = row < BN ? row : BN - 1;
    w_src[k] = (n0 + row) * w_row_len + ((cpos ^ ((row >> 2) & 3)) << 3);
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  auto issue = [&](int slot, int tap, int kc) {
    const int a_k = __builtin_amdgcn_readlane(tap_in_v, tap) + kc * BK;
    const int w_k = __builtin_amdgcn_readlane(tap_w_v, tap) + kc * BK;
    const unsigned dst0 = lds_base + slot * STAGE;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      glds16(in_base + (a_src[k] + a_k), __builtin_amdgcn_readfirstlane(dst0 + (wave + 4 * k) * 1024));
#pragma unroll
    for (int k = 0; k < IPW_W; ++k) {
      const int id = ((wave + 2) & 3) + 4 * k;
      if (id < W_INSTR) {  // wave-uniform
#ifdef NBDT_DMA_WTILED_FAKE   // timing experiment: what DMA-ordered weight tiles (one contiguous KiB per piece) would buy
        glds16(w_base + (((kc * ntaps + tap) * W_INSTR + id) * 512 + lane * 8),
               __builtin_amdgcn_readfirstlane(dst0 + A_BYTES + id * 1024));
#else
        glds16(w_base + (w_src[k] + w_k), __builtin_amdgcn_readfirstlane(dst0 + A_BYTES + id * 1024));
#endif
      }
    }
  };

  f32x16 acc[NT][2];
#pragma unroll
  for (int tn = 0; tn < NT; ++tn)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;

  const int frag_row = lane & 31;
  const int frag_half = lane >> 5;

  auto compute = [&](int slot) {
    const unsigned char* As = smem + slot * STAGE;
    const unsigned char* Ws = As + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 2 * ks + frag_half;
      bf16x8 pf[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
        pf[tm] = *(const bf16x8*)(As + lds_off(wave * 64 + tm * 32 + frag_row, c));
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const bf16x8 wf = *(const bf16x8*)(Ws + lds_off(tn * 32 + frag_row, c));
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, pf[tm], acc[tn][tm], 0, 0, 0);
      }
    }
  };

  // ---- pipeline over K tiles, tap fastest (see conv.hip for why)
  int tap = 0, kc = 0;
  auto advance = [&]() {
    if (++tap == ntaps) { tap = 0; ++kc; }
  };
  issue(0, tap, kc);
  advance();
  if (nk > 1) { issue(1, tap, kc); advance(); }
  int slot = 0;
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(MINPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < nk) {
      int s2 = slot + 2;
      s2 = s2 >= NSTAGE ? s2 - NSTAGE : s2;
      issue(s2, tap, kc);
      advance();
    }
    compute(slot);
    slot = slot + 1 == NSTAGE ? 0 : slot + 1;
  }

  static_assert(conv_epilogue_lds_bytes<NT>() <= NSTAGE * STAGE, "epilogue does not fit in the ring");
  conv_epilogue<NT, HAS_RES, STATS>(acc, p, epi_lds_packed<NT, 4>(smem, wave), m0, n0, m_blk, wave, lane, tid);
}

template <int NT, bool HAS_RES, int STATS>
__global__ __launch_bounds__(256, 2) void conv_igemm_dma_kernel(nbdt::ConvDmaParams p) {
  conv_igemm_dma_body<NT, HAS_RES, STATS>(p, blockIdx.x);
}

// Several launches of the kernel above in ONE grid: the four output-parity classes of a strided 3x3 data gradient
// (1 + 2 + 2 + 4 taps, disjoint output pixels, the same operands) are 128 pixel tiles each at 512 images -- a quarter
// of the chip per launch when they run one after the other.  Block b belongs to class c with first[c] <= b <
// first[c + 1] and runs that class's descriptor with the class-local block index (every first[] is a multiple of 8, so
// the XCD-contiguous item walk holds per class).
namespace nbdt {
struct ConvDmaMulti {
  ConvDmaParams p[4];
  int first[5];
};
}  // namespace nbdt
template <int NT, bool HAS_RES>
__global__ __launch_bounds__(256, 2) void conv_igemm_dma_multi_kernel(nbdt::ConvDmaMulti mp) {
  const int b = blockIdx.x;
  const int c = b >= mp.first[2] ? (b >= mp.first[3] ? 3 : 2) : (b >= mp.first[1] ? 1 : 0);
  conv_igemm_dma_body<NT, HAS_RES, 0>(mp.p[c], b - mp.first[c]);
}

namespace nbdt {

template <int NT>
static int launch_dma(ConvDmaParams& p, hipStream_t st) {
  constexpr int BN = 32 * NT;
  p.n_blocks = p.d.cout / BN;
  p.m_blocks = (p.M + BM - 1) / BM;
  const int items = p.m_blocks * p.n_blocks;
  p.per_xcd = (items + 7) / 8;
  const size_t shmem = (size_t)NSTAGE * ((BM * BK * 2) + (BN * BK * 2));
  static DeviceAttr site;     // one per NT instantiation
  if (site.need(shmem)) {
#define NBDT_ATTR(R, S)                                                                                         \
  NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_dma_kernel<NT, R, S>),    \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))
    NBDT_ATTR(true, 1); NBDT_ATTR(true, 0); NBDT_ATTR(false, 1); NBDT_ATTR(false, 0); NBDT_ATTR(false, 2);
    NBDT_ATTR(true, 3); NBDT_ATTR(false, 3);
#undef NBDT_ATTR
    site.done(shmem);
  }
  const dim3 grid(p.per_xcd * 8), blk(256);
#define NBDT_GO(R, S)                                                                                              \
  do {                                                                                                                \
    snprintf(g_last_igemm_full, sizeof(g_last_igemm_full), "conv_igemm_dma_kernel<%d, %s, %d>", NT, R ? "true" : "false", S); \
    hipLaunchKernelGGL((conv_igemm_dma_kernel<NT, R, S>), grid, blk, shmem, st, p);                                \
  } while (0)
  if (p.aff_scale != nullptr) { if (p.res != nullptr) NBDT_GO(true, 3); else NBDT_GO(false, 3); }
  else if (p.bn_x != nullptr) NBDT_GO(false, 2);
  else if (p.res != nullptr) { if (p.stats) NBDT_GO(true, 1); else NBDT_GO(true, 0); }
  else { if (p.stats) NBDT_GO(false, 1); else NBDT_GO(false, 0); }
#undef NBDT_GO
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

template <int NT>
static int launch_dma_multi(ConvDmaMulti& mp, int n, bool accumulate, hipStream_t st) {
  constexpr int BN = 32 * NT;
  int blocks = 0;
  for (int c = 0; c < 4; ++c) {
    mp.first[c] = blocks;
    if (c >= n) continue;
    ConvDmaParams& p = mp.p[c];
    p.n_blocks = p.d.cout / BN;
    p.m_blocks = (p.M + BM - 1) / BM;
    p.per_xcd = (p.m_blocks * p.n_blocks + 7) / 8;
    blocks += p.per_xcd * 8;
  }
  mp.first[4] = blocks;
  for (int c = n; c < 4; ++c) { mp.p[c] = mp.p[0]; mp.first[c] = blocks; }     // (never selected)
  const size_t shmem = (size_t)NSTAGE * ((BM * BK * 2) + (BN * BK * 2));
  static DeviceAttr site;     // one per NT instantiation
  if (site.need(shmem)) {
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_dma_multi_kernel<NT, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_dma_multi_kernel<NT, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    site.done(shmem);
  }
  snprintf(g_last_igemm_full, sizeof(g_last_igemm_full), "conv_igemm_dma_multi_kernel<%d, %s>", NT, accumulate ? "true" : "false");
  if (accumulate) hipLaunchKernelGGL((conv_igemm_dma_multi_kernel<NT, true>), dim3(blocks), dim3(256), shmem, st, mp);
  else hipLaunchKernelGGL((conv_igemm_dma_multi_kernel<NT, false>), dim3(blocks), dim3(256), shmem, st, mp);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

// called by nbdt_conv_igemm_multi (conv.hip) after argument validation: n <= 4 descriptors over the same operands
int conv_igemm_dma_multi(const nbdt_conv_desc* descs, int n, const void* in, const void* w, void* out,
                         hipStream_t st) {
  static thread_local ConvDmaMulti mp;      // (1.7 KB: kept off the stack of the ctypes caller)
  for (int c = 0; c < n; ++c) {
    ConvDmaParams& p = mp.p[c];
    p = ConvDmaParams{};
    p.d = descs[c];
    p.in = (const bf16_t*)in;
    p.w = (const bf16_t*)w;
    p.out = (bf16_t*)out;
    p.res = descs[c].accumulate ? (const bf16_t*)out : nullptr;
    p.M = descs[c].B * descs[c].gh * descs[c].gw;
    p.deterministic = deterministic() ? 1 : 0;
  }
  const bool acc = descs[0].accumulate != 0;
  const int nt32 = descs[0].cout / 32;
  if (nt32 % 5 == 0) return launch_dma_multi<5>(mp, n, acc, st);
  if (nt32 % 4 == 0) return launch_dma_multi<4>(mp, n, acc, st);
  if (nt32 % 3 == 0) return launch_dma_multi<3>(mp, n, acc, st);
  if (nt32 % 2 == 0) return launch_dma_multi<2>(mp, n, acc, st);
  return launch_dma_multi<1>(mp, n, acc, st);
}

// called by nbdt_conv_igemm (conv.hip) after argument validation
int conv_igemm_dma(const nbdt_conv_desc* d, const void* in, const void* w, void* out, const void* res,
                   float* stats, const BnBwdArgs* bn, int M, hipStream_t st) {
  ConvDmaParams p;
  p.d = *d;
  p.in = (const bf16_t*)in;
  p.w = (const bf16_t*)w;
  p.w_tiled = nullptr;
  p.out = (bf16_t*)out;
  p.res = (const bf16_t*)res;
  p.stats = stats;
  p.bn_x = bn ? (const bf16_t*)bn->x : nullptr;
  p.bn_mean = bn ? bn->mean : nullptr; p.bn_rstd = bn ? bn->rstd : nullptr;
  p.bn_gamma = bn ? bn->gamma : nullptr; p.bn_beta = bn ? bn->beta : nullptr;
  p.aff_scale = bn ? bn->aff_scale : nullptr; p.aff_shift = bn ? bn->aff_shift : nullptr;
  p.aff_act = bn ? bn->aff_act : 0;
  p.M = M;
  p.deterministic = deterministic() ? 1 : 0;
  const int nt32 = d->cout / 32;
  if (nt32 % 5 == 0) return launch_dma<5>(p, st);
  if (nt32 % 4 == 0) return launch_dma<4>(p, st);
  if (nt32 % 3 == 0) return launch_dma<3>(p, st);   // 96 / 192 / 672-wide MBConv projections
  if (nt32 % 2 == 0) return launch_dma<2>(p, st);
  return launch_dma<1>(p, st);
}

}  // namespace nbdt
