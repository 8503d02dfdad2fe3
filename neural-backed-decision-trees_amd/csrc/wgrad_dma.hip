// Weight-gradient kernel v3: LDS-DMA staging + hardware transpose reads (gfx950 only).
//
// Same contract as conv_wgrad_kernel (wgrad.hip) -- dw[co][tap][ci] += sum_m gy[m][co] * x[m+tap][ci]
// -- but the data path has NO staging registers and NO VALU transposes:
//
//   HBM --global_load_lds_dwordx4--> LDS tile [32 pixels][C channels] (natural NHWC order)
//       --ds_read_b64_tr_b16-->      MFMA operand registers (pixel-contiguous per lane)
//       --v_mfma_f32_16x16x32_bf16-> 80x80 fp32 accumulators per wave (160x160 per block)
//
//   * global_load_lds writes LDS at (wave-uniform base + lane*16): the LDS image of an operand is
//     a linear array of 16-byte pieces.  Piece order is [8-channel chunk][32 pixels]: one
//     wave-instruction moves 2 chunks x 32 pixels, lane -> (chunk = 2*instr + lane/32, position
//     lane%32).  Every DMA a lane issues in a stage therefore fetches the SAME pixel, so the
//     pixel -> (b,i,j) -> offset arithmetic is done once per operand per stage, not once per DMA
//     (v3.0 used pixel-major pieces: its address VALU work exceeded the MFMA time, 79M vs 59M
//     quad-cycles per launch).  Odd chunks store pixel (position ^ 8): the two chunks of a
//     16-channel operand block are 512 B apart (same banks), the XOR moves them to opposite
//     halves of the 256-byte bank row and the transpose reads below are conflict-free.
//   * ds_read_b64_tr_b16: each 16-lane group reads a [4 pixels][16 channels] block (lane t supplies
//     the address of row t>>2, 8-byte piece t&3) and lane t receives channel t's 4 pixels --
//     measured on MI355X by probes/layout_probe.hip.  Two reads give the 8 k-values of one
//     16x16x32 operand; k slot (g = lane>>4, j) maps to pixel 4g + (j&3) + 16*(j>>2) for BOTH
//     operands, so the contraction is over the same pixel in gy and x.
//   * 3-stage LDS ring, one raw s_barrier per 32-pixel stage, COUNTED s_waitcnt vmcnt: the DMA for
//     stage t+2 is issued right after the barrier that frees its slot and has two full MFMA
//     phases to land.  The DMA is issued from inline asm so hipcc's waitcnt insertion never sees
//     it (it would drain vmcnt(0) before every ds_read otherwise); all waits are placed by hand:
//         wait vmcnt(MINPW)  -> this wave's pieces of stage t have landed
//         s_barrier          -> everyone's pieces landed AND everyone finished reading stage t-1
//         issue DMA(t+2) into slot (t+2)%3 == (t-1)%3 ; transpose-read + 25 MFMAs on slot t%3
//   * K tail and idle lanes read the tensor's zero border pixel (padded layout), never garbage.
//
// Roofline: MFMA-bound, flops = 2*M*cout*cin*ntaps.
#include "common.h"
#include <stdlib.h>

using namespace nbdt;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace nbdt {

struct WgradDmaParams {
  nbdt_wgrad_desc d;
  const bf16_t* x;
  const bf16_t* gy;
  float* dw;
  long long dw_split_stride;   // 0, or (deterministic mode) elements of dw: a zeroed copy per pixel split
  int M;               // pixels
  int chunks;          // ceil(M / 32)
  int chunks_per_split;
  int n_ci_blocks;
  int splits, items, per_xcd;
  FastDiv div_gw, div_gh;
};

}  // namespace nbdt

constexpr int KS = 32;  // pixels per LDS stage
constexpr int NSTAGE = 3;

// LDS-DMA of 16 bytes per lane: LDS[lds_dst + lane*16 .. +16) <- *gsrc  (m0 = wave-uniform LDS base)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// wave w issues gy instructions {w, w+4, ..} and x instructions {(w+2)%4, (w+2)%4+4, ..}: the
// rotation balances the two remainders (10 + 10 instructions -> 5 per wave)
constexpr int min_dma_per_wave(int g_instr, int x_instr) {
  int best = 1 << 30;
  for (int w = 0; w < 4; ++w) {
    int n = 0;
    for (int id = w; id < g_instr; id += 4) ++n;
    for (int id = (w + 2) & 3; id < x_instr; id += 4) ++n;
    best = n < best ? n : best;
  }
  return best;
}

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma_kernel(nbdt::WgradDmaParams p) {
  constexpr int CG = 32 * WM, CX = 32 * WN;           // channels of the block tile (couts x cins)
  constexpr int G_BYTES = (CG / 8) * 512, X_BYTES = (CX / 8) * 512;   // [chunk][32 px][16 B]
  constexpr int STAGE = G_BYTES + X_BYTES;
  constexpr int G_INSTR = CG / 16, X_INSTR = CX / 16;  // wave-instructions per stage (2 chunks each)
  constexpr int IPG = (G_INSTR + 3) / 4, IPX = (X_INSTR + 3) / 4;  // DMA slots per wave per stage
  constexpr int MINPW = min_dma_per_wave(G_INSTR, X_INSTR);        // fewest DMAs any wave issues per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const nbdt_wgrad_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware work order, tap fastest (see wgrad.hip)
  const int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if (item >= p.items) return;
  const int tap = item % d.ntaps;
  const int rest = item / d.ntaps;
  const int split = rest % p.splits;
  const int tile = rest / p.splits;
  const int co_blk = tile / p.n_ci_blocks;
  const int ci_blk = tile - co_blk * p.n_ci_blocks;
  const int co0 = co_blk * CG;
  const int ci0 = ci_blk * CX;
  const int c_begin = split * p.chunks_per_split;
  int c_end = c_begin + p.chunks_per_split;
  c_end = c_end < p.chunks ? c_end : p.chunks;
  if (c_begin >= c_end) return;
  const int n_chunks = c_end - c_begin;

  // Everything the pipeline loop needs is pinned in SGPRs once (readfirstlane makes the value opaque:
  // the inline-asm DMA carries a "memory" clobber, and without this hipcc re-loads every kernel
  // argument with s_load + s_waitcnt lgkmcnt(0) in front of every DMA).
#define NBDT_PIN(x) __builtin_amdgcn_readfirstlane(x)
  const int M = NBDT_PIN(p.M);
  const int g_bs = NBDT_PIN(d.g_bs), g_hs = NBDT_PIN(d.g_hs), g_ws = NBDT_PIN(d.g_ws);
  const int x_bs = NBDT_PIN(d.x_bs), x_hs = NBDT_PIN(d.x_hs), x_ws = NBDT_PIN(d.x_ws);
  FastDiv dgw, dgh;
  dgw.mul = NBDT_PIN(p.div_gw.mul); dgw.sh = NBDT_PIN(p.div_gw.sh); dgw.d = NBDT_PIN(p.div_gw.d);
  dgh.mul = NBDT_PIN(p.div_gh.mul); dgh.sh = NBDT_PIN(p.div_gh.sh); dgh.d = NBDT_PIN(p.div_gh.d);
  const unsigned long long gy_u = (unsigned long long)p.gy, x_u = (unsigned long long)p.x;
  const bf16_t* gy_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(gy_u >> 32)) << 32) |
                                          (unsigned)NBDT_PIN((unsigned)gy_u));
  const bf16_t* x_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(x_u >> 32)) << 32) |
                                         (unsigned)NBDT_PIN((unsigned)x_u));
#undef NBDT_PIN

  // ---- DMA addressing.  Wave w issues gy wave-instructions {w, w+4, ..} and x instructions
  // {w, w+4, ..}; instruction `id` fills chunks 2*id (lanes 0-31) and 2*id+1 (lanes 32-63).
  const int hi_half = lane >> 5;
  const int my_px = (lane & 31) ^ (hi_half << 3);   // odd chunks hold pixel (position ^ 8)
  const int g_base_off = d.g_base + co0 + hi_half * 8;
  const int x_base_off = d.x_base + d.tap_off[tap] + ci0 + hi_half * 8;
  const int g_zero = co0 + hi_half * 8;             // top-left border pixel of the padded tensor: zeros
  const int x_zero = ci0 + hi_half * 8;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  auto issue = [&](int slot, int chunk) {
    const int m = chunk * KS + my_px;
    const bool valid = m < M;
    const unsigned mm = valid ? (unsigned)m : 0u;
    const unsigned t1 = fdiv(mm, dgw);
    const int j = (int)(mm - t1 * dgw.d);
    const unsigned b = fdiv(t1, dgh);
    const int i = (int)(t1 - b * dgh.d);
    const int og = valid ? ((int)b * g_bs + i * g_hs + j * g_ws + g_base_off) : g_zero;
    const int ox = valid ? ((int)b * x_bs + i * x_hs + j * x_ws + x_base_off) : x_zero;
    const bf16_t* gsrc = gy_base + og;
    const bf16_t* xsrc = x_base + ox;
    const unsigned dst0 = lds_base + slot * STAGE;
#pragma unroll
    for (int k = 0; k < IPG; ++k) {
      const int id = wave + 4 * k;
      if (id < G_INSTR)  // wave-uniform
        glds16(gsrc + id * 16, __builtin_amdgcn_readfirstlane(dst0 + id * 1024));
    }
#pragma unroll
    for (int k = 0; k < IPX; ++k) {
      const int id = ((wave + 2) & 3) + 4 * k;
      if (id < X_INSTR)
        glds16(xsrc + id * 16, __builtin_amdgcn_readfirstlane(dst0 + G_BYTES + id * 1024));
    }
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transpose-read addressing: lane t of 16-lane group g reads pixel row r = 4g + (t>>2) (+16 for
  // the second half), 8-byte piece (t&3) of a [32 px][16 channel] block = chunks (c0, c0+1):
  //   byte = (c0 + q) * 512 + ((r ^ 8q) * 16) + (t&1) * 8,   q = (t&3) >> 1
  const int g4 = lane >> 4, t16 = lane & 15;
  const int rr = 4 * g4 + (t16 >> 2);
  const int q = (t16 & 3) >> 1;
  const int lane_off = q * 512 + ((rr ^ (q << 3)) << 4) + (t16 & 1) * 8;
  const int g_lane_off = lane_off + wm * WM * 1024;   // 16 channels = 2 chunks = 1024 B
  const int x_lane_off = lane_off + wn * WN * 1024;

  auto frag = [&](const unsigned char* tile_lane, int ch_tile) -> bf16x8 {
    const unsigned char* a0 = tile_lane + ch_tile * 1024;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 256));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  auto compute = [&](int slot) {
    const unsigned char* Gs = smem + slot * STAGE + g_lane_off;
    const unsigned char* Xs = smem + slot * STAGE + G_BYTES + x_lane_off;
    bf16x8 gf[WM];
#pragma unroll
    for (int a = 0; a < WM; ++a) gf[a] = frag(Gs, a);
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const bf16x8 xf = frag(Xs, b);
#pragma unroll
      for (int a = 0; a < WM; ++a)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[a], xf, acc[a][b], 0, 0, 0);
    }
  };

  // ---- pipeline
  issue(0, c_begin);
  if (n_chunks > 1) issue(1, c_begin + 1);
  int slot = 0;
  for (int t = 0; t < n_chunks; ++t) {
    // this wave's DMA pieces of stage t have landed once at most the next stage's are outstanding
    if (t + 1 < n_chunks) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(MINPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    // (lgkmcnt(0): this wave's transpose reads of the slot about to be refilled have completed)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < n_chunks) {
      int s2 = slot + 2;
      s2 = s2 >= NSTAGE ? s2 - NSTAGE : s2;
      issue(s2, c_begin + t + 2);
    }
    compute(slot);
    slot = slot + 1 == NSTAGE ? 0 : slot + 1;
  }

  // ---- epilogue: 16x16 tile, lane: ci = col (lane&15), regs r -> co = 4*(lane>>4) + r
  const int w_tap = d.w_tap[tap];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const int ci = ci0 + (wn * WN + b) * 16 + t16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wm * WM + a) * 16 + 4 * g4 + r;
        atomicAdd(p.dw + (int64_t)split * p.dw_split_stride + ((int64_t)co * d.w_ntaps + w_tap) * d.cin + ci, acc[a][b][r]);
      }
    }
}

namespace nbdt {

template <int WM, int WN>
static int launch_dma(WgradDmaParams& p, hipStream_t st) {
  const nbdt_wgrad_desc& d = p.d;
  p.n_ci_blocks = d.cin / (32 * WN);
  const int tiles = d.ntaps * (d.cout / (32 * WM)) * p.n_ci_blocks;
  // 2 resident blocks per CU -> 512 slots per round; pick the split count that fills whole rounds
  // from BELOW (513 items on 512 slots runs two rounds for one block: measured -40%)
  int splits = 512 / tiles;
  const int max_splits = p.chunks / 16 > 0 ? p.chunks / 16 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.chunks_per_split = (p.chunks + splits - 1) / splits;
  splits = (p.chunks + p.chunks_per_split - 1) / p.chunks_per_split;
  p.splits = splits;
  p.items = tiles * splits;
  p.per_xcd = (p.items + 7) / 8;
  const size_t shmem = (size_t)NSTAGE * (32 * WM / 8 + 32 * WN / 8) * 512;
  static DeviceAttr site;     // one per (WM, WN) instantiation
  if (site.need(shmem)) {
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_kernel<WM, WN>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    site.done(shmem);
  }
  float* const dw = p.dw;
  const size_t dw_elems = (size_t)p.d.cout * p.d.w_ntaps * p.d.cin;
  p.dw_split_stride = 0;
  if (deterministic()) {
    float* rows = det_rows(st, (size_t)p.splits * dw_elems);
    if (!rows) return nbdt::fail(NBDT_ENOMEM, "deterministic mode: %s (%s)", "no workspace for the per-split gradients", nbdt::det_rows_why());
    NBDT_HIP_CHECK(hipMemsetAsync(rows, 0, (size_t)p.splits * dw_elems * sizeof(float), st));
    p.dw = rows;
    p.dw_split_stride = (long long)dw_elems;
  }
  hipLaunchKernelGGL((conv_wgrad_dma_kernel<WM, WN>), dim3(p.per_xcd * 8), dim3(256), shmem, st, p);
  NBDT_LAUNCH_CHECK();
  if (p.dw != dw) return det_fold(st, p.dw, p.splits, dw_elems, dw);
  return NBDT_OK;
}

// called by nbdt_conv_wgrad (wgrad.hip) after argument validation
int wgrad_dma(const nbdt_wgrad_desc* d, const void* x, const void* gy, float* dw, hipStream_t st) {
  WgradDmaParams p;
  p.d = *d;
  p.x = (const bf16_t*)x;
  p.gy = (const bf16_t*)gy;
  p.dw = dw;
  p.M = d->B * d->gh * d->gw;
  p.chunks = (p.M + KS - 1) / KS;
  p.div_gw = make_fastdiv((unsigned)d->gw);
  p.div_gh = make_fastdiv((unsigned)d->gh);
  const int mt = d->cout / 32, nt = d->cin / 32;
#define NBDT_WG(WM_, WN_) return launch_dma<WM_, WN_>(p, st)
#define NBDT_WG_ROW(WM_)                 \
  do {                                   \
    if (nt % 5 == 0) NBDT_WG(WM_, 5);    \
    if (nt % 4 == 0) NBDT_WG(WM_, 4);    \
    if (nt % 3 == 0) NBDT_WG(WM_, 3);    \
    if (nt % 2 == 0) NBDT_WG(WM_, 2);    \
    NBDT_WG(WM_, 1);                     \
  } while (0)
  if (mt % 5 == 0) NBDT_WG_ROW(5);
  if (mt % 4 == 0) NBDT_WG_ROW(4);
  if (mt % 3 == 0) NBDT_WG_ROW(3);
  if (mt % 2 == 0) NBDT_WG_ROW(2);
  NBDT_WG_ROW(1);
#undef NBDT_WG_ROW
#undef NBDT_WG
}

}  // namespace nbdt
