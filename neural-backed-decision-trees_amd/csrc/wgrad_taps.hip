// Weight-gradient kernel v4 for dense 3x3 / stride-1 convs: ALL NINE TAPS in one block (gfx950 only).
//
// wgrad_dma.hip (v3) gives each (tap, cout-tile, cin-tile) its own block, so every tap re-streams both the
// gy tile and the (shifted) x tile through L2: 20 KB per 32-pixel stage for 1.6 MFLOP (80 flop/B -- the
// same L2-bandwidth wall the forward kernel hit before its halo tile).  Here a block owns
// (32*WM couts) x 32 cins x 9 taps: per stage it loads the gy tile ONCE and the x tile ONCE with its
// one-pixel halo ((RS+2) x (CS+2) pixels), and the nine taps are nine shifted transpose-reads of the same
// LDS tile: 18 KB per 2.95 MFLOP (164 flop/B), 45 MFMAs per barrier instead of 25.
//
//   waves: 2 (cout halves) x 2 (16-cin halves); per wave acc[9 taps][WM] of v_mfma_f32_16x16x32_bf16
//   LDS stage: gy [chunk(8 co)][32 px][16 B] + x [chunk(8 ci)][128 halo slots][16 B]; odd chunks store
//   slot^8 (same bank argument as wgrad_dma.hip); ring of 3 stages, counted vmcnt, one raw barrier per stage.
//   A stage is 32 consecutive pixels = an RS x CS rectangle of one image (RS = max(1,32/W), CS = min(W,32)),
//   so every DMA address is (wave-uniform stage base) + (per-lane constant): one add per DMA.
// Partial sums over the pixel splits are combined with fp32 atomics into dw (+= semantics).
#include "common.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

using namespace nbdt;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

namespace nbdt {
struct WgradTapsParams {
  nbdt_wgrad_desc d;
  const bf16_t* x;
  const bf16_t* gy;
  float* dw;
  long long dw_split_stride;   // 0: every pixel split adds into dw; deterministic mode: elements of dw, a zeroed
                               // copy per split (one add per address; det_fold sums the copies in split order)
  int stages;            // M / 32
  int stages_per_split;
  int n_ci_blocks;       // cin / 32
  int splits, items, per_xcd;
  int store;                   // K-split kernel: 1 = plain stores into this split's copy (every (split, tile) block writes
                               // its whole tile, so the copies need no zeroing); det_fold sums the copies into dw
  int rs, cs;            // stage rectangle (rows x cols), rs*cs == 32
  int hw2;               // cs + 2
  int hp;                // (rs+2)*(cs+2) halo pixels (<= 128)
  int stages_per_row;    // gw / cs   (>= 1)
  int rowgroups;         // gh / rs
  FastDiv div_spr, div_rg;
};
}  // namespace nbdt

constexpr int KS = 32;
#ifndef NBDT_WGT_NSTAGE
#define NBDT_WGT_NSTAGE 3
#endif
constexpr int NSTAGE = NBDT_WGT_NSTAGE;   // LDS ring depth; stages are prefetched NSTAGE-1 ahead
constexpr int XSLOTS = 128;                    // halo slots per ci chunk (hp <= 102 used)
// x tile: (CX/8) chunks x 128 slots x 16 B = 8 KiB for 32 cins (4 waves), 16 KiB for 64 cins (8 waves)
constexpr int x_bytes(int nwv) { return (nwv * 8 / 8) * XSLOTS * 16; }

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

// DMA instructions: gy = 2*WM ids (2 chunks x 32 px each), x = 2*NWV ids (chunk = id>>1, slots 64*(id&1)..+64).
// Wave w issues gy ids {w, w+NWV, ..} and x ids {w, w+NWV}: x is always 2 per wave.
constexpr int min_g_dma(int g_instr, int nwv) {
  int best = 1 << 30;
  for (int w = 0; w < nwv; ++w) {
    int n = 0;
    for (int id = w; id < g_instr; id += nwv) ++n;
    best = n < best ? n : best;
  }
  return best;
}

// NWV = 4: block = 32*WM couts x 32 cins, 2 blocks per CU.  NWV = 8: 32*WM couts x 64 cins, 1 block per CU --
// the same gy tile feeds twice the cins.  Ablation (160->160 @32x32, B=512): MFMA+LDS without DMA 193 us, the
// DMA stream alone 253 us, everything 295 us: the kernel is bound by the 1.5 GB it pulls through L2 into LDS
// (4x the tensors' size: gy is re-read once per cin block, the x halo is 3.2x its core), so bytes per MFMA
// are the lever; 64 cins per block cut them by 28 %.
template <int WM, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void conv_wgrad_taps_kernel(nbdt::WgradTapsParams p) {
  constexpr int CG = 32 * WM;
  constexpr int NWN = NWV / 2;                 // 16-cin groups (waves along cin)
  constexpr int CX = 16 * NWN;
  constexpr int X_BYTES = x_bytes(NWV);
  constexpr int G_BYTES = (CG / 8) * 512;
  constexpr int STAGE = G_BYTES + X_BYTES;
  constexpr int G_INSTR = CG / 16;
  constexpr int IPG = (G_INSTR + NWV - 1) / NWV;
  constexpr int MINPW = min_g_dma(G_INSTR, NWV) + 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const nbdt_wgrad_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;

  const int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if (item >= p.items) return;
  // tile fastest: the blocks that share one pixel range (same split, different cout/cin tile) are neighbours in
  // item order, i.e. co-resident on ONE XCD, so its L2 fetches their common gy / x tiles from HBM once.
  // (split-fastest order spread them over the 8 XCDs: TCC hit rate 27 %, 1.4 GB of HBM reads for 0.36 GB
  // of tensors.)
  const int n_tiles = p.items / p.splits;
  const int tile = item % n_tiles;
  const int split = item / n_tiles;
  const int co_blk = tile / p.n_ci_blocks;
  const int ci_blk = tile - co_blk * p.n_ci_blocks;
  const int co0 = co_blk * CG;
  const int ci0 = ci_blk * CX;
  const int s_begin = split * p.stages_per_split;
  int s_end = s_begin + p.stages_per_split;
  s_end = s_end < p.stages ? s_end : p.stages;
  if (s_begin >= s_end) return;
  const int n_st = s_end - s_begin;

#define NBDT_PIN(x) __builtin_amdgcn_readfirstlane(x)
  const int g_bs = NBDT_PIN(d.g_bs), g_hs = NBDT_PIN(d.g_hs), g_ws = NBDT_PIN(d.g_ws);
  const int x_bs = NBDT_PIN(d.x_bs), x_hs = NBDT_PIN(d.x_hs), x_ws = NBDT_PIN(d.x_ws);
  const int rs = NBDT_PIN(p.rs), cs = NBDT_PIN(p.cs), hw2 = NBDT_PIN(p.hw2), hp_n = NBDT_PIN(p.hp);
  FastDiv dspr, drg;
  dspr.mul = NBDT_PIN(p.div_spr.mul); dspr.sh = NBDT_PIN(p.div_spr.sh); dspr.d = NBDT_PIN(p.div_spr.d);
  drg.mul = NBDT_PIN(p.div_rg.mul); drg.sh = NBDT_PIN(p.div_rg.sh); drg.d = NBDT_PIN(p.div_rg.d);
  const unsigned long long gy_u = (unsigned long long)p.gy, x_u = (unsigned long long)p.x;
  const bf16_t* gy_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(gy_u >> 32)) << 32) |
                                          (unsigned)NBDT_PIN((unsigned)gy_u));
  const bf16_t* x_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(x_u >> 32)) << 32) |
                                         (unsigned)NBDT_PIN((unsigned)x_u));
#undef NBDT_PIN

  // ---- per-lane DMA constants
  const int hi_half = lane >> 5;
  // gy: instruction id fills chunks 2*id (lanes 0-31) and 2*id+1 (lanes 32-63); odd chunk stores px^8
  const int g_px = (lane & 31) ^ (hi_half << 3);
  const int g_lane_src = (g_px / cs) * g_hs + (g_px % cs) * g_ws + d.g_base + co0 + hi_half * 8;
  // x: instruction id -> chunk id>>1 (8 cins), halo slots 64*(id&1) + lane; odd chunks store slot^8.
  // wave w issues ids w and w+NWV: slot half (w&1) for both
  int x_lane_src[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int id = wave + NWV * k;
    const int chunk = id >> 1;
    int slot = 64 * (id & 1) + lane;
    int hp = (chunk & 1) ? (slot ^ 8) : slot;
    hp = hp < hp_n ? hp : hp_n - 1;           // unused slots re-fetch the last halo pixel
    x_lane_src[k] = (hp / hw2) * x_hs + (hp % hw2) * x_ws + d.x_base + ci0 + chunk * 8;
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  auto issue = [&](int slot_i, int stage) {
    // stage -> (image b, first row r0, first col c0): wave-uniform
    const unsigned st = (unsigned)stage;
    const unsigned q1 = fdiv(st, dspr);                 // st / stages_per_row
    const int sc = (int)(st - q1 * dspr.d);             // stage column index within the row group
    const unsigned b = fdiv(q1, drg);                   // / rowgroups
    const int rg = (int)(q1 - b * drg.d);
    const int r0 = rg * rs, c0 = sc * cs;
    const int g_stage = (int)b * g_bs + r0 * g_hs + c0 * g_ws;
    const int x_stage = (int)b * x_bs + r0 * x_hs + c0 * x_ws;
    const bf16_t* gsrc = gy_base + (g_stage + g_lane_src);
    const unsigned dst0 = lds_base + slot_i * STAGE;
#pragma unroll
    for (int k = 0; k < IPG; ++k) {
      const int id = wave + NWV * k;
      if (id < G_INSTR)
        glds16(gsrc + id * 16, __builtin_amdgcn_readfirstlane(dst0 + id * 1024));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int id = wave + NWV * k;
      glds16(x_base + (x_stage + x_lane_src[k]), __builtin_amdgcn_readfirstlane(dst0 + G_BYTES + id * 1024));
    }
  };

  f32x4 acc[9][WM];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < WM; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- transpose-read addressing (see wgrad_dma.hip): lane t of group g: k rows 4g + (t>>2) (+16),
  // 8-byte piece t&3 -> chunk q = (t&3)>>1, half t&1
  const int g4 = lane >> 4, t16 = lane & 15;
  const int rr = 4 * g4 + (t16 >> 2);
  const int q = (t16 & 3) >> 1;
  const int g_lane_off = q * 512 + ((rr ^ (q << 3)) << 4) + (t16 & 1) * 8 + wm * WM * 1024;
  // x: halo slot of output pixel k at tap (0,0): (k / cs) * hw2 + k % cs
  int xpos[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = rr + 16 * h;
    xpos[h] = (k / cs) * hw2 + (k % cs);
  }
  const int x_lane_off = (wn * 2 + q) * (XSLOTS * 16) + (t16 & 1) * 8;
  const int qx = q << 3;

  auto compute = [&](int slot_i) {
    const unsigned char* Gs = smem + slot_i * STAGE + g_lane_off;
    const unsigned char* Xs = smem + slot_i * STAGE + G_BYTES + x_lane_off;
    bf16x8 gf[WM];
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const unsigned char* a0 = Gs + a * 1024;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 256));
      gf[a] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int toff = (t / 3) * hw2 + (t % 3);
      const unsigned char* p0 = Xs + (((xpos[0] + toff) ^ qx) << 4);
      const unsigned char* p1 = Xs + (((xpos[1] + toff) ^ qx) << 4);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
      const bf16x8 xf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int a = 0; a < WM; ++a)
        acc[t][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[a], xf, acc[t][a], 0, 0, 0);
    }
  };

  // ---- pipeline (same protocol as wgrad_dma.hip)
  constexpr int PD = NSTAGE - 1;
#pragma unroll
  for (int i = 0; i < PD; ++i)
    if (i < n_st) issue(i, s_begin + i);
  int slot_i = 0;
  for (int t = 0; t < n_st; ++t) {
    // stages t+1 .. t+PD-1 may stay in flight (fewer near the end)
    int after = n_st - 1 - t;
    after = after < PD - 1 ? after : PD - 1;
    if (after >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * MINPW) : "memory");
    else if (after == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(MINPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + PD < n_st) {
      int s2 = slot_i + PD;
      s2 = s2 >= NSTAGE ? s2 - NSTAGE : s2;
      issue(s2, s_begin + t + PD);
    }
    compute(slot_i);
    slot_i = slot_i + 1 == NSTAGE ? 0 : slot_i + 1;
  }

  // ---- epilogue: acc[tap][a][r]: co = co0 + (wm*WM + a)*16 + 4*g4 + r ; ci = ci0 + wn*16 + t16
  const int ci = ci0 + wn * 16 + t16;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int w_tap = d.w_tap[t];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wm * WM + a) * 16 + 4 * g4 + r;
        atomicAdd(p.dw + (int64_t)split * p.dw_split_stride + ((int64_t)co * d.w_ntaps + w_tap) * d.cin + ci, acc[t][a][r]);
      }
  }
}


// ------------------------------------------------------------------------------------------------------------
// 8-wave ping-pong form (1 block per CU), built like conv3x3_pp_kernel (conv_halo.hip).  Block tile as above --
// (32*WM couts) x 32 cins x 9 taps -- but the nine taps are split between two wave groups that work on the SAME
// LDS stage one barrier apart: waves 0-3 own taps 0-4, waves 4-7 taps 5-8 (inside a group: 2 cout halves x 2 cin
// halves).  So each stage is loaded once for 8 waves (half the DMA per MFMA of two 4-wave blocks; the 4-wave
// kernel keeps TA 65 % busy at 42 % MFMA utilisation), a wave holds 100 accumulator registers instead of 180, and
// its step splits into a load segment (20 transpose reads -> registers, its share of the DMA of stage u+4) and an
// MFMA segment (25 or 20 MFMAs, plus the address arithmetic of the next load segment in their shadow):
//     group 0:  bP  L0 b M0 b  L1 b M1 b ...            group 1:  bP b  L0 b M0 b  L1 b M1 ...
// Stages are 64 pixels (two 32-pixel MFMA K chunks).  Ring of 5 stage slots: stage u+3 is issued in L(u) into the
// slot stage u-2 left (its last reader, group 1's M(u-2), finished before the barrier in front of group 0's
// L(u-1)); a wave retires what it issued two load segments ago (vmcnt: one stage's worth may stay in flight)
// before the barrier that precedes the first read.
#ifndef NBDT_WPP_TIMING
#define NBDT_WPP_TIMING 0   // 1: s_memtime stamps around the segments, per-wave sums in g_wpp_timing (scratch/wpp_timing.py)
#endif
#if NBDT_WPP_TIMING
__device__ unsigned g_wpp_timing[2048 * 8];     // [block*8 + wave][8]: load segment, barrier 1, MFMA segment, barrier 2, total, stages
extern "C" int nbdt_debug_wpp_timing(unsigned* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wpp_timing), sizeof(unsigned) * 2048 * 8);
}
__device__ __forceinline__ unsigned wstamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return (unsigned)t;
}
#define NBDT_WSTAMP(acc_) { const unsigned t_ = wstamp(); acc_ += t_ - tm_prev; tm_prev = t_; }
#else
#define NBDT_WSTAMP(acc_)
#endif
template <int WM>
__global__ __launch_bounds__(512, 2) void conv_wgrad_pp_kernel(nbdt::WgradTapsParams p) {
  constexpr int CG = 32 * WM;
  constexpr int KSP = 64, KK = KSP / 32;       // pixels per stage: two 32-pixel MFMA K chunks
  constexpr int PG = 2 * CG;                   // gy row pitch in bytes
  constexpr int XS = 144;                      // x halo slots per stage (64 B each): 9 pieces
  constexpr int G_BYTES = KSP * PG;
  constexpr int X_BYTES = XS * 64;
  constexpr int STAGE = G_BYTES + X_BYTES;
  constexpr int G_INSTR = G_BYTES / 1024;      // 4 * WM gy pieces per stage
  constexpr int X_INSTR = XS / 16;
  constexpr int IPG = (G_INSTR + 7) / 8;
  constexpr int NSLOT = 5, PD = 3;
  constexpr int NT0 = 5;                       // taps of group 0 (group 1: 9 - NT0)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const nbdt_wgrad_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;

  const int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if (item >= p.items) return;
  const int n_tiles = p.items / p.splits;      // tile fastest: blocks sharing a pixel range sit on one XCD
  const int tile = item % n_tiles;
  const int split = item / n_tiles;
  const int co_blk = tile / p.n_ci_blocks;
  const int ci_blk = tile - co_blk * p.n_ci_blocks;
  const int co0 = co_blk * CG;
  const int ci0 = ci_blk * 32;
  const int s_begin = split * p.stages_per_split;
  int s_end = s_begin + p.stages_per_split;
  s_end = s_end < p.stages ? s_end : p.stages;
  if (s_begin >= s_end) return;
#ifdef NBDT_WPP_FRAC8        // timing experiment (scratch/variants): only FRAC8/8 of the stages -- a kernel that much faster
  const int n_st = (s_end - s_begin) * NBDT_WPP_FRAC8 / 8;
#else
  const int n_st = s_end - s_begin;
#endif

#define NBDT_PIN(x) __builtin_amdgcn_readfirstlane(x)
  const int g_bs = NBDT_PIN(d.g_bs), g_hs = NBDT_PIN(d.g_hs), g_ws = NBDT_PIN(d.g_ws);
  const int x_bs = NBDT_PIN(d.x_bs), x_hs = NBDT_PIN(d.x_hs), x_ws = NBDT_PIN(d.x_ws);
  const int rs = NBDT_PIN(p.rs), cs = NBDT_PIN(p.cs), hw2 = NBDT_PIN(p.hw2), hp_n = NBDT_PIN(p.hp);
  FastDiv dspr, drg;
  dspr.mul = NBDT_PIN(p.div_spr.mul); dspr.sh = NBDT_PIN(p.div_spr.sh); dspr.d = NBDT_PIN(p.div_spr.d);
  drg.mul = NBDT_PIN(p.div_rg.mul); drg.sh = NBDT_PIN(p.div_rg.sh); drg.d = NBDT_PIN(p.div_rg.d);
  const unsigned long long gy_u = (unsigned long long)p.gy, x_u = (unsigned long long)p.x;
  const bf16_t* gy_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(gy_u >> 32)) << 32) |
                                          (unsigned)NBDT_PIN((unsigned)gy_u));
  const bf16_t* x_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(x_u >> 32)) << 32) |
                                         (unsigned)NBDT_PIN((unsigned)x_u));
#undef NBDT_PIN

  // ---- LDS stage image: PIXEL-MAJOR.  gy [64 px][CG couts] (2*CG-byte rows), x [144 halo slots][32 cins] (64-byte
  // rows).  The chunk-major image of the 4-wave kernel makes a DMA instruction touch 64 (x) or 32 (gy) cache lines
  // for its 1 KiB, and the texture path retires ~0.45 lines per clock whatever their payload (probes/dma_probe.hip:
  // 7.2 and 14.4 B/clk/CU against 52 for contiguous KiB) -- per 32-pixel stage that is 1850 cycles of address
  // processing for 765 cycles of MFMA: the 4-wave kernel's 42 % MFMA utilisation IS that ratio.  Pixel-major rows
  // make a gy piece 3.2 whole pixel rows (8-9 lines) and an x piece 16 pixels x 64 B (16 lines).
  // ds_read_b64_tr_b16 gathers [4 px][16 ch] blocks with a free row stride, so it reads this image directly.
  //
  // K order (round 5).  Which pixel an MFMA k index stands for is free as long as both operands agree.  The 16-lane
  // group g4 of a fragment holds k = 8 g4 .. 8 g4 + 7; rounds 2-4 gave it pixels {4 g4 .. +3, 16 + 4 g4 .. +3} of the
  // 32-pixel chunk, now it has EIGHT CONSECUTIVE pixels 8 g4 .. 8 g4 + 7 of one image row (8 | cs).  Then the x operands
  // of the three taps of a kernel row (r, 0..2) are the windows [c, c+8), [c+1, c+9), [c+2, c+10) of ONE 12-pixel run
  // of halo row y + r: three transpose reads (pixels c..c+3, c+4..c+7, c+8..c+11) serve all three taps -- tap s = 0 and
  // s = 2 are register sub-ranges, s = 1 is four v_perm -- where each tap used to read its own two blocks: 6 x-reads
  // per K chunk and wave instead of 10 / 8 (16 reads per chunk with the 10 of gy, was 20 / 18).  A wave's 8-byte
  // transpose reads complete one per ~38 cycles whatever surrounds them (s_memtime, profiles/r05_wgrad_segments.txt:
  // 16 reads 610 cycles of a 970-cycle load segment), so reads per MFMA are a lever: load segment 1022 -> 972 cycles,
  // MFMA segment of group 0 (reads of the second chunk in its shadow) 1053 -> 977, stage 2236 -> 2123.
  // Bank swizzle: the two 16-lane groups of a 32-lane LDS pass read pixels 8 apart (or, for 8-wide images, one image row
  // apart), i.e. the same 64-byte rows of the 256-byte bank line; the 32-byte half a group reads is XOR-swapped by a key
  // that differs between the two -- gy: bit 3 of the pixel; x: (halo column >> 3) ^ halo row -- on the DMA source
  // address and on the read address.
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned g_voff[IPG];                        // gy pieces {wave + 8k}: piece = LDS bytes [1024 id, +1024)
#pragma unroll
  for (int k = 0; k < IPG; ++k) {
    const int pos = (wave + 8 * k) * 1024 + lane * 16;
    int px = pos / PG;
    const int j = (pos - px * PG) >> 4;        // 16-byte chunk of the row this lane fills
    const int src_chunk = j ^ (((px >> 3) & 1) << 1);
    px = px < KSP ? px : KSP - 1;              // (ids past the tile are never issued)
    g_voff[k] = (unsigned)((px / cs) * g_hs + (px % cs) * g_ws + d.g_base + co0 + src_chunk * 8) * 2u;
  }
  unsigned x_voff[2];                          // x pieces {wave, 8 (wave 0 only)}: halo slots [16 id, +16) x 64 B
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int slot = 16 * (wave + 8 * k) + (lane >> 2);
    const int hp = slot < hp_n ? slot : hp_n - 1;          // unused slots re-fetch the last halo pixel
    const int hrow = hp / hw2, hcol = hp - hrow * hw2;
    const int src_chunk = (lane & 3) ^ ((((hcol >> 3) ^ hrow) & 1) << 1);
    x_voff[k] = (unsigned)(hrow * x_hs + hcol * x_ws + d.x_base + ci0 + src_chunk * 8) * 2u;
  }
  int n_mine = 1 + (wave + 8 < X_INSTR ? 1 : 0);           // DMA instructions this wave issues per stage
#pragma unroll
  for (int k = 0; k < IPG; ++k) n_mine += (wave + 8 * k < G_INSTR) ? 1 : 0;

  // stage -> scalar element offsets of its gy / x tiles (image b, first row r0, first col c0)
  auto stage_off = [&](int stage, int& g_stage, int& x_stage) {
    const unsigned st = (unsigned)stage;       // (branch-free divisions: a branch would split the MFMA segment)
    const unsigned q1m = __umulhi(st, dspr.mul) >> dspr.sh;
    const unsigned q1 = dspr.d == 1 ? st : q1m;
    const int sc = (int)(st - q1 * dspr.d);
    const unsigned bm = __umulhi(q1, drg.mul) >> drg.sh;
    const unsigned b = drg.d == 1 ? q1 : bm;
    const int rg = (int)(q1 - b * drg.d);
    const int r0 = rg * rs, c0 = sc * cs;
    g_stage = (int)b * g_bs + r0 * g_hs + c0 * g_ws;
    x_stage = (int)b * x_bs + r0 * x_hs + c0 * x_ws;
  };
  auto issue = [&](int slot_i, int g_stage, int x_stage) {
    const unsigned dst0 = lds_base + slot_i * STAGE;
#pragma unroll
    for (int k = 0; k < IPG; ++k)
      if (wave + 8 * k < G_INSTR) glds16_s(gy_base + g_stage, g_voff[k], dst0 + (wave + 8 * k) * 1024);
    glds16_s(x_base + x_stage, x_voff[0], dst0 + G_BYTES + wave * 1024);
    if (wave + 8 < X_INSTR) glds16_s(x_base + x_stage, x_voff[1], dst0 + G_BYTES + (wave + 8) * 1024);
  };

  f32x4 acc[NT0][WM];
#pragma unroll
  for (int t = 0; t < NT0; ++t)
#pragma unroll
    for (int a = 0; a < WM; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- transpose-read addressing: 16-lane group g4 reads pixels 8 g4 .. 8 g4+3 (and +4) of a 32-pixel K chunk,
  // lane t16 the 8 bytes (4 channels) number t16&3 of pixel 8 g4 + (t16>>2) in a 16-channel block; it ends up with
  // channel t16 of the four pixels.
  typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;
  typedef __attribute__((address_space(3))) s16x4* lds_tr;
  const lds_cptr smem3 = (lds_cptr)smem;
  const int g4 = lane >> 4, t16 = lane & 15;
  const int pr = t16 >> 2;                     // pixel of the 4-pixel block this lane addresses
  const int c8 = (t16 & 3) * 8;
  // gy: 16-cout block b of pixel px sits at px*PG + (b ^ ((px>>3)&1))*32; (px>>3)&1 == g4&1 for px = 32 kk + 8 g4 + j.
  // Blocks b and b+2 swap the same way: two per-lane bases (tiles a = 0, 1), the rest are +64 B immediates.
  int g_lane_off[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) g_lane_off[a] = (8 * g4 + pr) * PG + (((wm * WM + a) ^ (g4 & 1)) * 32) + c8;
  // x: halo row / column (tap (0,0)) of pixel 32 kk + 8 g4 of the stage rectangle, the start of this group's run
  int xrow0[KK], xcol0[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int k = kk * 32 + 8 * g4;
    xrow0[kk] = k / cs;
    xcol0[kk] = k - xrow0[kk] * cs;
  }
  const int x_lane_off = G_BYTES + c8;

  // ---- prologue: stages 0 .. PD-1, all landed before the first load segment
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (u < n_st) {
      int gs, xs;
      stage_off(s_begin + u, gs, xs);
      issue(u, gs, xs);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                // bP
  if (grp == 1) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // One group's main loop; T0 / NTP are literals so every register array is statically indexed.
  auto run = [&](auto t0_c, auto ntp_c) {
    constexpr int T0 = decltype(t0_c)::value, NTP = decltype(ntp_c)::value;
    // x fragment offsets inside a stage slot: per lane, per (K chunk, kernel row, 4-pixel block of the 12-pixel run) --
    // stage independent, so they are computed once; a stage only adds its slot base (one v_add each, in the MFMA
    // segment's shadow).  This group's taps T0 .. T0+NTP-1 lie in kernel rows R0 .. R0+NR-1.
    constexpr int R0 = T0 / 3, NR = (T0 + NTP - 1) / 3 - R0 + 1;
    int xrel[KK][NR][3];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int hrow = xrow0[kk] + R0 + r, hcol = xcol0[kk] + 4 * q + pr;
          const int hs = hrow * hw2 + hcol;                // halo slot; its 32-byte half: wn ^ key(row, column)
          xrel[kk][r][q] = x_lane_off + ((hs << 6) | (((((hcol >> 3) ^ hrow) ^ wn) & 1) << 5));
        }
    lds_cptr xa[KK][NR][3];                    // x block addresses of the next load segment
    lds_cptr ga[2];
    auto prepare = [&](int slot_i) {
      int off = slot_i * STAGE;
      asm volatile("" : "+s"(off));            // the adds stay in the segment that calls this
      const lds_cptr base = smem3 + off;
      ga[0] = base + g_lane_off[0];
      ga[1] = base + g_lane_off[1];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
          for (int q = 0; q < 3; ++q) xa[kk][r][q] = base + xrel[kk][r][q];
    };
    prepare(0);
    int slot_n = 1 % NSLOT, slot_d = PD % NSLOT;
    int g_next = 0, x_next = 0;                // tile offsets of stage u + PD, computed in M(u-1)
    if (PD < n_st) stage_off(s_begin + PD, g_next, x_next);
#if NBDT_WPP_TIMING
    unsigned tm_l = 0, tm_b1 = 0, tm_m = 0, tm_b2 = 0;
    const unsigned tm_begin = wstamp();
    unsigned tm_prev = tm_begin;
#endif
    for (int u = 0; u < n_st; ++u) {
      // ================= L(u): fragments -> registers, this wave's DMA pieces of stage u + PD =================
      bf16x8 gf[KK][WM];
      u32x2 xw[KK][NR][3];                     // the 12-pixel runs: [block q] = pixels 4q .. 4q+3 of this lane's channel
      auto read_chunk = [&](auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          const lds_cptr a0 = ga[a & 1] + ((a >> 1) * 64 + kk * 32 * PG);
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0 + 4 * PG));
          gf[kk][a] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            xw[kk][r][q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)xa[kk][r][q]));
      };
      // x operand of tap T0 + t for K chunk kk: pixels s .. s+7 of the run of kernel row (T0 + t) / 3
      auto x_frag = [&](int kk, int t) {
        const int r = (T0 + t) / 3 - R0, sft = (T0 + t) % 3;
        const u32x2 A = xw[kk][r][0], B = xw[kk][r][1], C = xw[kk][r][2];
        u32x4_t v;
        if (sft == 0) v = u32x4_t{A[0], A[1], B[0], B[1]};
        else if (sft == 2) v = u32x4_t{A[1], B[0], B[1], C[0]};
        else v = u32x4_t{__builtin_amdgcn_alignbit(A[1], A[0], 16), __builtin_amdgcn_alignbit(B[0], A[1], 16),
                         __builtin_amdgcn_alignbit(B[1], B[0], 16), __builtin_amdgcn_alignbit(C[0], B[1], 16)};
        return __builtin_bit_cast(bf16x8, v);
      };
      // Only the FIRST K chunk's fragments are read here: 8-byte LDS reads reach their rate only with several waves
      // per SIMD in flight, and a load segment has one -- all 40 reads took ~870 cycles (s_memtime), longer than
      // the partner's MFMA segment.  The second chunk is read inside the MFMA segment, under the first chunk's MFMAs.
#if NBDT_WPP_TIMING == 3     // diagnosis only: DMA wait first, so that its stamp does not include the LDS reads
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      NBDT_WSTAMP(tm_m)       // (timing 3: column "mfma-seg" = vmcnt wait alone)
#endif
      read_chunk(std::integral_constant<int, 0>{});
      // stage u+1 (issued two load segments ago) must be in LDS before the next barrier; the stage issued since
      // may stay in flight.  Near the end nothing was issued: drain.
      if (u + PD <= n_st) {                    // the previous load segment issued a full stage
        switch (n_mine) {
#define NBDT_CASE(K) case K: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
          NBDT_CASE(1) NBDT_CASE(2) NBDT_CASE(3) NBDT_CASE(4) NBDT_CASE(5)
#undef NBDT_CASE
          default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#if NBDT_WPP_TIMING > 1
      NBDT_WSTAMP(tm_b1)      // (fine timing: column "barrier1" = reads issued + vmcnt wait)
#endif

      if (u + PD < n_st) issue(slot_d, g_next, x_next);
#if NBDT_WPP_TIMING > 1
      NBDT_WSTAMP(tm_b2)      // (fine timing: column "barrier2" = DMA issue)
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      NBDT_WSTAMP(tm_l)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#if NBDT_WPP_TIMING == 1
      NBDT_WSTAMP(tm_b1)
#elif NBDT_WPP_TIMING == 3
      NBDT_WSTAMP(tm_l)
#else
      NBDT_WSTAMP(tm_m)
#endif
      // ================= M(u): KK * NTP * WM MFMAs; their shadow prepares L(u+1) =================
      __builtin_amdgcn_s_setprio(1);
      read_chunk(std::integral_constant<int, 1>{});
      prepare(slot_n);
      stage_off(s_begin + u + 1 + PD, g_next, x_next);     // (past the end: computed, never used)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int t = 0; t < NTP; ++t)
#pragma unroll
          for (int a = 0; a < WM; ++a)
            acc[t][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[kk][a], x_frag(kk, t), acc[t][a], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < KK * NTP * WM; ++i) {              // one MFMA, one of each other kind in its shadow
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // VALU
        __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);   // SALU
      }
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int r = 0; r < NR; ++r) asm volatile("" : "+v"(xa[kk][r][0]), "+v"(xa[kk][r][1]), "+v"(xa[kk][r][2]));
      asm volatile("" : "+v"(ga[0]), "+v"(ga[1]));
      __builtin_amdgcn_s_setprio(0);
      NBDT_WSTAMP(tm_m)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#if NBDT_WPP_TIMING == 1
      NBDT_WSTAMP(tm_b2)
#elif NBDT_WPP_TIMING == 3
      NBDT_WSTAMP(tm_l)
#else
      NBDT_WSTAMP(tm_m)
#endif
      slot_n = slot_n + 1 == NSLOT ? 0 : slot_n + 1;
      slot_d = slot_d + 1 == NSLOT ? 0 : slot_d + 1;
    }
#if NBDT_WPP_TIMING
    if (lane == 0 && item < 256) {
      unsigned* o = g_wpp_timing + (item * 8 + wave) * 8;
      o[0] = tm_l; o[1] = tm_b1; o[2] = tm_m; o[3] = tm_b2; o[4] = tm_prev - tm_begin; o[5] = n_st;
    }
#endif
    // ---- epilogue: acc[t][a][r]: co = co0 + (wm*WM + a)*16 + 4*g4 + r ; ci = ci0 + wn*16 + t16
    if (grp == 0) __builtin_amdgcn_s_barrier();
    const int ci = ci0 + wn * 16 + t16;
#ifdef NBDT_WPP_NO_EPI       // timing experiment: what the fp32 atomics of the epilogue cost (accumulators stay live)
    float keep = 0.f;
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) keep += acc[t][a][r];
    if (keep == 12345.678f) p.dw[ci] = keep;
#else
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      const int w_tap = d.w_tap[T0 + t];
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + (wm * WM + a) * 16 + 4 * g4 + r;
          atomicAdd(p.dw + (int64_t)split * p.dw_split_stride + ((int64_t)co * d.w_ntaps + w_tap) * d.cin + ci, acc[t][a][r]);
        }
    }
#endif
  };
  if (grp == 0) run(std::integral_constant<int, 0>{}, std::integral_constant<int, NT0>{});
  else run(std::integral_constant<int, NT0>{}, std::integral_constant<int, 9 - NT0>{});
}

// ------------------------------------------------------------------------------------------------------------
// K-split form (round 5): the two wave groups split the PIXELS of a stage, not the taps.
//
// What bounds conv_wgrad_pp_kernel (profiles/r05_wgrad_segments.txt, r05_wgrad_pp3_ab.txt): a CU completes one
// ds_read_b64_tr_b16 per ~8 cycles however many waves issue them, and a stage of the tap-split kernel needs 8 waves x 32
// reads = 256 of them = 2050 cycles for 1440 cycles of matrix pipe -- the measured 2120.  (The 12-wave, one-kernel-row-per-
// group kernel below does 312 reads per stage and is 22 % slower: exactly the ratio.)  The redundancy is in the tiling: a
// gy fragment is read by the two groups AND the two cin halves, four times per stage.
// Here a wave owns 80 couts x 16 cins x ALL NINE taps (180 accumulator registers) and the groups take the two 32-pixel
// K chunks of a stage: group 0 chunk 0, group 1 chunk 1, one barrier apart as before.  A wave reads its gy fragments once
// for nine taps (10 reads) and one 12-pixel run per kernel row (9 reads): 8 x 19 = 152 reads per stage, 1220 cycles,
// under the 1440 of the pipe.  All reads are in the load segment; the MFMA segment is 45 MFMAs and address arithmetic.
// The two groups' partial sums meet at the end: through LDS, lane to lane with the partner wave (w ^ 4), in two rounds
// (taps 0-4 to group 0, taps 5-8 to group 1: 102 + 82 KB of the ring, which is dead by then), so that the fp32 atomics
// that follow are the same 100 / 80 per lane as before -- the split count, i.e. the atomic volume, does not change.
// Ring, prefetch distance, DMA pieces, swizzles and the vmcnt protocol are conv_wgrad_pp_kernel's.
template <int WM>
__global__ __launch_bounds__(512, 2) void conv_wgrad_ks_kernel(nbdt::WgradTapsParams p) {
  constexpr int CG = 32 * WM;
  constexpr int KSP = 64;
  constexpr int PG = 2 * CG;
  constexpr int XS = 144;
  constexpr int G_BYTES = KSP * PG;
  constexpr int X_BYTES = XS * 64;
  constexpr int STAGE = G_BYTES + X_BYTES;
  constexpr int G_INSTR = G_BYTES / 1024;
  constexpr int X_INSTR = XS / 16;
  constexpr int IPG = (G_INSTR + 7) / 8;
  constexpr int NSLOT = 5, PD = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const nbdt_wgrad_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;    // group = K chunk of every stage
  const int wm = w4 >> 1, wn = w4 & 1;

  const int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if (item >= p.items) return;
  const int n_tiles = p.items / p.splits;
  const int tile = item % n_tiles;
  const int split = item / n_tiles;
  const int co_blk = tile / p.n_ci_blocks;
  const int ci_blk = tile - co_blk * p.n_ci_blocks;
  const int co0 = co_blk * CG;
  const int ci0 = ci_blk * 32;
  const int s_begin = split * p.stages_per_split;
  int s_end = s_begin + p.stages_per_split;
  s_end = s_end < p.stages ? s_end : p.stages;
  if (s_begin >= s_end) return;
  const int n_st = s_end - s_begin;

#define NBDT_PIN(x) __builtin_amdgcn_readfirstlane(x)
  const int g_bs = NBDT_PIN(d.g_bs), g_hs = NBDT_PIN(d.g_hs), g_ws = NBDT_PIN(d.g_ws);
  const int x_bs = NBDT_PIN(d.x_bs), x_hs = NBDT_PIN(d.x_hs), x_ws = NBDT_PIN(d.x_ws);
  const int rs = NBDT_PIN(p.rs), cs = NBDT_PIN(p.cs), hw2 = NBDT_PIN(p.hw2), hp_n = NBDT_PIN(p.hp);
  FastDiv dspr, drg;
  dspr.mul = NBDT_PIN(p.div_spr.mul); dspr.sh = NBDT_PIN(p.div_spr.sh); dspr.d = NBDT_PIN(p.div_spr.d);
  drg.mul = NBDT_PIN(p.div_rg.mul); drg.sh = NBDT_PIN(p.div_rg.sh); drg.d = NBDT_PIN(p.div_rg.d);
  const unsigned long long gy_u = (unsigned long long)p.gy, x_u = (unsigned long long)p.x;
  const bf16_t* gy_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(gy_u >> 32)) << 32) |
                                          (unsigned)NBDT_PIN((unsigned)gy_u));
  const bf16_t* x_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(x_u >> 32)) << 32) |
                                         (unsigned)NBDT_PIN((unsigned)x_u));
#undef NBDT_PIN

  // ---- LDS stage image, DMA pieces: conv_wgrad_pp_kernel's
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned g_voff[IPG];
#pragma unroll
  for (int k = 0; k < IPG; ++k) {
    const int pos = (wave + 8 * k) * 1024 + lane * 16;
    int px = pos / PG;
    const int j = (pos - px * PG) >> 4;
    const int src_chunk = j ^ (((px >> 3) & 1) << 1);
    px = px < KSP ? px : KSP - 1;
    g_voff[k] = (unsigned)((px / cs) * g_hs + (px % cs) * g_ws + d.g_base + co0 + src_chunk * 8) * 2u;
  }
  unsigned x_voff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int slot = 16 * (wave + 8 * k) + (lane >> 2);
    const int hp = slot < hp_n ? slot : hp_n - 1;
    const int hrow = hp / hw2, hcol = hp - hrow * hw2;
    const int src_chunk = (lane & 3) ^ ((((hcol >> 3) ^ hrow) & 1) << 1);
    x_voff[k] = (unsigned)(hrow * x_hs + hcol * x_ws + d.x_base + ci0 + src_chunk * 8) * 2u;
  }
  int n_mine = 1 + (wave + 8 < X_INSTR ? 1 : 0);
#pragma unroll
  for (int k = 0; k < IPG; ++k) n_mine += (wave + 8 * k < G_INSTR) ? 1 : 0;

  auto stage_off = [&](int stage, int& g_stage, int& x_stage) {
    const unsigned st = (unsigned)stage;
    const unsigned q1m = __umulhi(st, dspr.mul) >> dspr.sh;
    const unsigned q1 = dspr.d == 1 ? st : q1m;
    const int sc = (int)(st - q1 * dspr.d);
    const unsigned bm = __umulhi(q1, drg.mul) >> drg.sh;
    const unsigned b = drg.d == 1 ? q1 : bm;
    const int rg = (int)(q1 - b * drg.d);
    const int r0 = rg * rs, c0 = sc * cs;
    g_stage = (int)b * g_bs + r0 * g_hs + c0 * g_ws;
    x_stage = (int)b * x_bs + r0 * x_hs + c0 * x_ws;
  };
  auto issue = [&](int slot_i, int g_stage, int x_stage) {
    const unsigned dst0 = lds_base + slot_i * STAGE;
#pragma unroll
    for (int k = 0; k < IPG; ++k)
      if (wave + 8 * k < G_INSTR) glds16_s(gy_base + g_stage, g_voff[k], dst0 + (wave + 8 * k) * 1024);
    glds16_s(x_base + x_stage, x_voff[0], dst0 + G_BYTES + wave * 1024);
    if (wave + 8 < X_INSTR) glds16_s(x_base + x_stage, x_voff[1], dst0 + G_BYTES + (wave + 8) * 1024);
  };

  f32x4 acc[9][WM];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < WM; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- transpose-read addressing: this group's K chunk = pixels 32 grp .. 32 grp + 31 of every stage
  typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;
  typedef __attribute__((address_space(3))) s16x4* lds_tr;
  const lds_cptr smem3 = (lds_cptr)smem;
  const int g4 = lane >> 4, t16 = lane & 15;
  const int pr = t16 >> 2;
  const int c8 = (t16 & 3) * 8;
  int g_lane_off[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
    g_lane_off[a] = (grp * 32 + 8 * g4 + pr) * PG + (((wm * WM + a) ^ (g4 & 1)) * 32) + c8;   // ((32 grp + 8 g4 + j) >> 3) & 1 == g4 & 1
  int xrel[3][3];                              // [kernel row][4-pixel block of the 12-pixel run]
  {
    const int k = grp * 32 + 8 * g4;
    const int row0 = k / cs, col0 = k - row0 * cs;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int hrow = row0 + r, hcol = col0 + 4 * q + pr;
        const int hs = hrow * hw2 + hcol;
        xrel[r][q] = G_BYTES + c8 + ((hs << 6) | (((((hcol >> 3) ^ hrow) ^ wn) & 1) << 5));
      }
  }

  // ---- prologue
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (u < n_st) {
      int gs, xs;
      stage_off(s_begin + u, gs, xs);
      issue(u, gs, xs);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                // bP
  if (grp == 1) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  lds_cptr xa[3][3];
  lds_cptr ga[2];
  auto prepare = [&](int slot_i) {
    int off = slot_i * STAGE;
    asm volatile("" : "+s"(off));
    const lds_cptr base = smem3 + off;
    ga[0] = base + g_lane_off[0];
    ga[1] = base + g_lane_off[1];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) xa[r][q] = base + xrel[r][q];
  };
  prepare(0);
  int slot_n = 1 % NSLOT, slot_d = PD % NSLOT;
  int g_next = 0, x_next = 0;
  if (PD < n_st) stage_off(s_begin + PD, g_next, x_next);

  for (int u = 0; u < n_st; ++u) {
    // ================= L(u): this wave's fragments of its chunk -> registers, its pieces of stage u + PD =================
    bf16x8 gf[WM];
    u32x2 xw[3][3];
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const lds_cptr a0 = ga[a & 1] + (a >> 1) * 64;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0 + 4 * PG));
      gf[a] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        xw[r][q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)xa[r][q]));
    // stage u+1 (issued two load segments ago) must be in LDS before the next barrier; the stage issued since may fly
    if (u + PD <= n_st) {
      switch (n_mine) {
#define NBDT_CASE(K) case K: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
        NBDT_CASE(1) NBDT_CASE(2) NBDT_CASE(3) NBDT_CASE(4) NBDT_CASE(5)
#undef NBDT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifndef NBDT_WKS_DMA_IN_M
    if (u + PD < n_st) issue(slot_d, g_next, x_next);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= M(u): 9 x WM MFMAs; their shadow prepares L(u+1) =================
#ifdef NBDT_WKS_DMA_IN_M      // experiment: this wave's pieces at the head of its MFMA segment instead of in the load segment
    if (u + PD < n_st) issue(slot_d, g_next, x_next);
    __builtin_amdgcn_sched_barrier(0);
#endif
    __builtin_amdgcn_s_setprio(1);
    prepare(slot_n);
    stage_off(s_begin + u + 1 + PD, g_next, x_next);       // (past the end: computed, never used)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int r = t / 3, sft = t % 3;
      const u32x2 A = xw[r][0], B = xw[r][1], C = xw[r][2];
      u32x4_t v;
      if (sft == 0) v = u32x4_t{A[0], A[1], B[0], B[1]};
      else if (sft == 2) v = u32x4_t{A[1], B[0], B[1], C[0]};
      else v = u32x4_t{__builtin_amdgcn_alignbit(A[1], A[0], 16), __builtin_amdgcn_alignbit(B[0], A[1], 16),
                       __builtin_amdgcn_alignbit(B[1], B[0], 16), __builtin_amdgcn_alignbit(C[0], B[1], 16)};
      const bf16x8 xf = __builtin_bit_cast(bf16x8, v);
#pragma unroll
      for (int a = 0; a < WM; ++a)
        acc[t][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[a], xf, acc[t][a], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 9 * WM; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // VALU
      __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);   // SALU
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) asm volatile("" : "+v"(xa[r][0]), "+v"(xa[r][1]), "+v"(xa[r][2]));
    asm volatile("" : "+v"(ga[0]), "+v"(ga[1]));
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    slot_n = slot_n + 1 == NSLOT ? 0 : slot_n + 1;
    slot_d = slot_d + 1 == NSLOT ? 0 : slot_d + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  // ---- the two groups' partial sums meet: partner wave w ^ 4 holds the same (cout, cin) tile of the other K chunks.
  // Lane to lane through LDS ([f32x4 index][lane] x 16 B: lane-linear, conflict-free), two rounds; the ring is dead (every
  // wave is past its last read and no LDS-DMA is in flight: the last load segments drained vmcnt).
  __builtin_amdgcn_sched_barrier(0);
#ifndef NBDT_WKS_NO_EXCHANGE
  {
    typedef __attribute__((address_space(3))) f32x4* lds_f4;
    const lds_f4 ex = (lds_f4)(__attribute__((address_space(3))) unsigned char*)smem + (w4 * (5 * WM) * 64 + lane);
    if (grp == 1) {
#pragma unroll
      for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int a = 0; a < WM; ++a) ex[(t * WM + a) * 64] = acc[t][a];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[t][a] += ex[(t * WM + a) * 64];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int t = 5; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < WM; ++a) ex[((t - 5) * WM + a) * 64] = acc[t][a];
    }
    __syncthreads();
    if (grp == 1) {
#pragma unroll
      for (int t = 5; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[t][a] += ex[((t - 5) * WM + a) * 64];
    }
  }
#endif
  // ---- epilogue: group 0 owns taps 0-4, group 1 taps 5-8; co = co0 + (wm*WM + a)*16 + 4*g4 + r ; ci = ci0 + wn*16 + t16
  const int ci = ci0 + wn * 16 + t16;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#ifndef NBDT_WKS_NO_EXCHANGE                   // (debug build: no exchange, both groups add all their partial sums)
    if ((t < 5) != (grp == 0)) continue;       // wave-uniform
#endif
    const int w_tap = d.w_tap[t];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wm * WM + a) * 16 + 4 * g4 + r;
        float* const dst = p.dw + (int64_t)split * p.dw_split_stride + ((int64_t)co * d.w_ntaps + w_tap) * d.cin + ci;
        if (p.store) *dst = acc[t][a][r];
        else atomicAdd(dst, acc[t][a][r]);
      }
  }
}

// ------------------------------------------------------------------------------------------------------------
// 12-wave form (round 5): THREE wave groups, one kernel ROW each (taps 3g, 3g+1, 3g+2), three waves per SIMD.
//
// Why.  s_memtime stamps of the 8-wave kernel above (profiles/r05_wgrad_segments.txt): a wave's ds_read_b64_tr_b16
// complete one per ~38 cycles whatever surrounds them (16 reads: 610 cycles), its 3-5 LDS-DMA pieces cost 270 cycles of
// issue, and both sit in the load segment (970 cycles) while the partner's MFMA segment is 850 / 680 cycles of matrix
// pipe: the per-WAVE serial work (32 reads + 45 MFMAs + pieces per stage = 2200 cycles) sets the 2120-cycle stage, not
// the pipe (1440).  With one kernel row per wave the x operand of its three taps is ONE 12-pixel run (3 reads per K
// chunk), so a wave has 26 reads + 30 MFMAs + 1-3 pieces per stage (~1630 cycles), and with three waves per SIMD two of
// them are in load phases while the third issues MFMAs:
//     group g:   [g idle slots]  P1(0) b P2(0) b M(0) b  P1(1) b P2(1) b M(1) b ...   [2 - g idle slots]
//   P1(u): 13 transpose reads of K chunk 0 of stage u | this wave's pieces of stage u+3 | vmcnt: stage u+1 landed
//   P2(u): 13 transpose reads of K chunk 1
//   M(u):  30 MFMAs (no LDS access), the address arithmetic of stage u+1 in their shadow
// Every slot ends in ONE s_barrier executed by all 12 waves; in any slot exactly one group is in each phase.
// Hazards.  Stage u+3 overwrites the ring slot of stage u-2 (5 slots); its last reader is group 2's P2(u-2) in slot
// 3u-3, drained (lgkmcnt) before that slot's barrier; the first writer is group 0's P1(u) in slot 3u.  A wave retires
// (vmcnt) in P1(v) what it issued in P1(v-2) -- stage v+1, first read in slot 3v+3 by group 0 -- and P1(v) of every
// group lies in slots 3v .. 3v+2, a barrier before.  Stages alive in slot 3u: u-1 (group 2 reading), u, u+1, u+2 and
// u+3 arriving: the five slots.
template <int WM>
__global__ __launch_bounds__(768, 3) void conv_wgrad_pp3_kernel(nbdt::WgradTapsParams p) {
  constexpr int CG = 32 * WM;
  constexpr int KSP = 64, KK = KSP / 32;
  constexpr int PG = 2 * CG;
  constexpr int XS = 144;
  constexpr int G_BYTES = KSP * PG;
  constexpr int X_BYTES = XS * 64;
  constexpr int STAGE = G_BYTES + X_BYTES;
  constexpr int G_INSTR = G_BYTES / 1024;      // 4 * WM gy pieces per stage
  constexpr int X_INSTR = XS / 16;             // 9
  constexpr int NWV = 12;
  constexpr int IPG = (G_INSTR + NWV - 1) / NWV;
  constexpr int NSLOT = 5, PD = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const nbdt_wgrad_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;    // group = kernel row
  const int wm = w4 >> 1, wn = w4 & 1;

  const int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if (item >= p.items) return;
  const int n_tiles = p.items / p.splits;      // tile fastest: blocks sharing a pixel range sit on one XCD
  const int tile = item % n_tiles;
  const int split = item / n_tiles;
  const int co_blk = tile / p.n_ci_blocks;
  const int ci_blk = tile - co_blk * p.n_ci_blocks;
  const int co0 = co_blk * CG;
  const int ci0 = ci_blk * 32;
  const int s_begin = split * p.stages_per_split;
  int s_end = s_begin + p.stages_per_split;
  s_end = s_end < p.stages ? s_end : p.stages;
  if (s_begin >= s_end) return;
  const int n_st = s_end - s_begin;

#define NBDT_PIN(x) __builtin_amdgcn_readfirstlane(x)
  const int g_bs = NBDT_PIN(d.g_bs), g_hs = NBDT_PIN(d.g_hs), g_ws = NBDT_PIN(d.g_ws);
  const int x_bs = NBDT_PIN(d.x_bs), x_hs = NBDT_PIN(d.x_hs), x_ws = NBDT_PIN(d.x_ws);
  const int rs = NBDT_PIN(p.rs), cs = NBDT_PIN(p.cs), hw2 = NBDT_PIN(p.hw2), hp_n = NBDT_PIN(p.hp);
  FastDiv dspr, drg;
  dspr.mul = NBDT_PIN(p.div_spr.mul); dspr.sh = NBDT_PIN(p.div_spr.sh); dspr.d = NBDT_PIN(p.div_spr.d);
  drg.mul = NBDT_PIN(p.div_rg.mul); drg.sh = NBDT_PIN(p.div_rg.sh); drg.d = NBDT_PIN(p.div_rg.d);
  const unsigned long long gy_u = (unsigned long long)p.gy, x_u = (unsigned long long)p.x;
  const bf16_t* gy_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(gy_u >> 32)) << 32) |
                                          (unsigned)NBDT_PIN((unsigned)gy_u));
  const bf16_t* x_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(x_u >> 32)) << 32) |
                                         (unsigned)NBDT_PIN((unsigned)x_u));
#undef NBDT_PIN

  // ---- LDS stage image and its swizzles: exactly conv_wgrad_pp_kernel's (gy [64 px][CG], x [144 slots][64 B])
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned g_voff[IPG];                        // gy pieces {wave + 12 k}
#pragma unroll
  for (int k = 0; k < IPG; ++k) {
    const int pos = (wave + NWV * k) * 1024 + lane * 16;
    int px = pos / PG;
    const int j = (pos - px * PG) >> 4;
    const int src_chunk = j ^ (((px >> 3) & 1) << 1);
    px = px < KSP ? px : KSP - 1;              // (ids past the tile are never issued)
    g_voff[k] = (unsigned)((px / cs) * g_hs + (px % cs) * g_ws + d.g_base + co0 + src_chunk * 8) * 2u;
  }
  unsigned x_voff;                             // x piece {wave} (waves 0 .. 8): halo slots [16 wave, +16) x 64 B
  {
    const int slot = 16 * wave + (lane >> 2);
    const int hp = slot < hp_n ? slot : hp_n - 1;          // unused slots re-fetch the last halo pixel
    const int hrow = hp / hw2, hcol = hp - hrow * hw2;
    const int src_chunk = (lane & 3) ^ ((((hcol >> 3) ^ hrow) & 1) << 1);
    x_voff = (unsigned)(hrow * x_hs + hcol * x_ws + d.x_base + ci0 + src_chunk * 8) * 2u;
  }
  int n_mine = wave < X_INSTR ? 1 : 0;         // DMA instructions this wave issues per stage (1 .. 3 at WM = 5)
#pragma unroll
  for (int k = 0; k < IPG; ++k) n_mine += (wave + NWV * k < G_INSTR) ? 1 : 0;

  auto stage_off = [&](int stage, int& g_stage, int& x_stage) {
    const unsigned st = (unsigned)stage;       // (branch-free divisions: a branch would split the MFMA phase)
    const unsigned q1m = __umulhi(st, dspr.mul) >> dspr.sh;
    const unsigned q1 = dspr.d == 1 ? st : q1m;
    const int sc = (int)(st - q1 * dspr.d);
    const unsigned bm = __umulhi(q1, drg.mul) >> drg.sh;
    const unsigned b = drg.d == 1 ? q1 : bm;
    const int rg = (int)(q1 - b * drg.d);
    const int r0 = rg * rs, c0 = sc * cs;
    g_stage = (int)b * g_bs + r0 * g_hs + c0 * g_ws;
    x_stage = (int)b * x_bs + r0 * x_hs + c0 * x_ws;
  };
  auto issue = [&](int slot_i, int g_stage, int x_stage) {
    const unsigned dst0 = lds_base + slot_i * STAGE;
#pragma unroll
    for (int k = 0; k < IPG; ++k)
      if (wave + NWV * k < G_INSTR) glds16_s(gy_base + g_stage, g_voff[k], dst0 + (wave + NWV * k) * 1024);
    if (wave < X_INSTR) glds16_s(x_base + x_stage, x_voff, dst0 + G_BYTES + wave * 1024);
  };

  f32x4 acc[3][WM];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int a = 0; a < WM; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- transpose-read addressing (conv_wgrad_pp_kernel's K order: 8 consecutive pixels per 16-lane group)
  typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;
  typedef __attribute__((address_space(3))) s16x4* lds_tr;
  const lds_cptr smem3 = (lds_cptr)smem;
  const int g4 = lane >> 4, t16 = lane & 15;
  const int pr = t16 >> 2;
  const int c8 = (t16 & 3) * 8;
  int g_lane_off[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) g_lane_off[a] = (8 * g4 + pr) * PG + (((wm * WM + a) ^ (g4 & 1)) * 32) + c8;
  int xrel[KK][3];                             // this group's kernel row: halo row = pixel row + grp
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int k = kk * 32 + 8 * g4;
    const int row0 = k / cs, col0 = k - row0 * cs;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int hrow = row0 + grp, hcol = col0 + 4 * q + pr;
      const int hs = hrow * hw2 + hcol;
      xrel[kk][q] = G_BYTES + c8 + ((hs << 6) | (((((hcol >> 3) ^ hrow) ^ wn) & 1) << 5));
    }
  }

  // ---- prologue: stages 0 .. PD-1, all landed before the first read
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (u < n_st) {
      int gs, xs;
      stage_off(s_begin + u, gs, xs);
      issue(u, gs, xs);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                // bP
  for (int i = 0; i < grp; ++i) __builtin_amdgcn_s_barrier();     // this group's idle slots at the start
  __builtin_amdgcn_sched_barrier(0);

  lds_cptr xa[KK][3];
  lds_cptr ga[2];
  auto prepare = [&](int slot_i) {
    int off = slot_i * STAGE;
    asm volatile("" : "+s"(off));
    const lds_cptr base = smem3 + off;
    ga[0] = base + g_lane_off[0];
    ga[1] = base + g_lane_off[1];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int q = 0; q < 3; ++q) xa[kk][q] = base + xrel[kk][q];
  };
  prepare(0);
  int slot_n = 1 % NSLOT, slot_d = PD % NSLOT;
  int g_next = 0, x_next = 0;                  // tile offsets of stage u + PD, computed in M(u-1)
  if (PD < n_st) stage_off(s_begin + PD, g_next, x_next);

  for (int u = 0; u < n_st; ++u) {
    bf16x8 gf[KK][WM];
    u32x2 xw[KK][3];
    auto read_chunk = [&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value;
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const lds_cptr a0 = ga[a & 1] + ((a >> 1) * 64 + kk * 32 * PG);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0 + 4 * PG));
        gf[kk][a] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int q = 0; q < 3; ++q)
        xw[kk][q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)xa[kk][q]));
    };
    auto x_frag = [&](int kk, int sft) {       // pixels sft .. sft + 7 of the run
      const u32x2 A = xw[kk][0], B = xw[kk][1], C = xw[kk][2];
      u32x4_t v;
      if (sft == 0) v = u32x4_t{A[0], A[1], B[0], B[1]};
      else if (sft == 2) v = u32x4_t{A[1], B[0], B[1], C[0]};
      else v = u32x4_t{__builtin_amdgcn_alignbit(A[1], A[0], 16), __builtin_amdgcn_alignbit(B[0], A[1], 16),
                       __builtin_amdgcn_alignbit(B[1], B[0], 16), __builtin_amdgcn_alignbit(C[0], B[1], 16)};
      return __builtin_bit_cast(bf16x8, v);
    };
    // ================= P1(u): K chunk 0 -> registers, this wave's pieces of stage u + PD =================
    const bool dma_now = u + PD < n_st;
    read_chunk(std::integral_constant<int, 0>{});
    if (dma_now) issue(slot_d, g_next, x_next);
    {
      // stage u+1 (issued in P1(u-2)) must be in LDS before this slot's barrier; what P1(u-1) and this phase issued
      // (stages u+2, u+3: n_mine pieces each, fewer near the end) may stay in flight
      const int fly = (dma_now ? n_mine : 0) + (u + PD - 1 < n_st ? n_mine : 0);
      switch (fly) {
#define NBDT_CASE(K) case K: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
        NBDT_CASE(1) NBDT_CASE(2) NBDT_CASE(3) NBDT_CASE(4) NBDT_CASE(5) NBDT_CASE(6)
#undef NBDT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= P2(u): K chunk 1 -> registers =================
    read_chunk(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= M(u): 2 x 3 x WM MFMAs; their shadow prepares P1(u+1) =================
    __builtin_amdgcn_s_setprio(1);
    prepare(slot_n);
    stage_off(s_begin + u + 1 + PD, g_next, x_next);       // (past the end: computed, never used)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < WM; ++a)
          acc[t][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[kk][a], x_frag(kk, t), acc[t][a], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < KK * 3 * WM; ++i) {                // one MFMA, one VALU / SALU in its shadow
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // VALU
      __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);   // SALU
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) asm volatile("" : "+v"(xa[kk][0]), "+v"(xa[kk][1]), "+v"(xa[kk][2]));
    asm volatile("" : "+v"(ga[0]), "+v"(ga[1]));
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    slot_n = slot_n + 1 == NSLOT ? 0 : slot_n + 1;
    slot_d = slot_d + 1 == NSLOT ? 0 : slot_d + 1;
  }
  for (int i = grp; i < 2; ++i) __builtin_amdgcn_s_barrier();     // this group's idle slots at the end
  // ---- epilogue: acc[t][a][r]: tap 3 grp + t; co = co0 + (wm*WM + a)*16 + 4*g4 + r ; ci = ci0 + wn*16 + t16
  const int ci = ci0 + wn * 16 + t16;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int w_tap = d.w_tap[3 * grp + t];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wm * WM + a) * 16 + 4 * g4 + r;
        atomicAdd(p.dw + (int64_t)split * p.dw_split_stride + ((int64_t)co * d.w_ntaps + w_tap) * d.cin + ci, acc[t][a][r]);
      }
  }
}

namespace nbdt {

static bool stage_geometry(const nbdt_wgrad_desc* d, int ks, int max_hp, WgradTapsParams* p);

bool wgrad_taps_applicable(const nbdt_wgrad_desc* d) {
  if (d->ntaps != 9 || d->x_base != 0) return false;
  if (d->x_ws != d->cin || d->x_hs != (d->gw + 2) * d->cin || d->x_bs != (d->gh + 2) * d->x_hs) return false;
  for (int t = 0; t < 9; ++t)
    if (d->tap_off[t] != (t / 3) * d->x_hs + (t % 3) * d->x_ws) return false;
  return stage_geometry(d, 32, 128, nullptr);
}

// Stage rectangle (rs x cs pixels of one image) and halo for stages of `ks` pixels; false when they do not tile it.
static bool stage_geometry(const nbdt_wgrad_desc* d, int ks, int max_hp, WgradTapsParams* p) {
  const int gw = d->gw, gh = d->gh;
  if (!(gw % ks == 0 || ks % gw == 0)) return false;
  int rs = gw >= ks ? 1 : ks / gw;
  int cs = gw >= ks ? ks : gw;
  if (ks == 64 && gw >= 64 && gw % 32 == 0 && gh % 2 == 0 && (rs + 2) * (cs + 2) > max_hp) {
    // rows of 64+ pixels: a 64-pixel stage as one row has a 3 x 66 halo (198 slots, the 8-wave kernels hold 144); as a
    // 2 x 32 rectangle -- two stages per 64 columns of a row pair -- it is 4 x 34 = 136 (round 5: ResNet18 at 64x64)
    rs = 2;
    cs = 32;
  }
  if (gh % rs != 0) return false;
  if ((rs + 2) * (cs + 2) > max_hp) return false;
  const long long M = (long long)d->B * gh * gw;
  if (!p) return true;
  p->stages = (int)(M / ks);
  p->rs = rs; p->cs = cs;
  p->hw2 = cs + 2;
  p->hp = (rs + 2) * (cs + 2);
  p->stages_per_row = gw / cs;
  p->rowgroups = gh / rs;
  p->div_spr = make_fastdiv((unsigned)p->stages_per_row);
  p->div_rg = make_fastdiv((unsigned)p->rowgroups);
  return true;
}

// PP: the 8-wave ping-pong kernel, 64-pixel stages, 1 block per CU (256 block slots); else 4-wave blocks,
// 32-pixel stages, 2 per CU (512 slots)
// Pixel split of a launch: fills whole rounds of resident blocks from BELOW (513 items on 512 slots cost 40 %);
// d.cu_budget: the caller wants only that many CUs filled (an HBM-bound pass on another stream gets the rest).
static void split_items(WgradTapsParams& p, int wm, bool pp) {
  const nbdt_wgrad_desc& d = p.d;
  p.n_ci_blocks = d.cin / 32;
  const int tiles = (d.cout / (32 * wm)) * p.n_ci_blocks;
  int cus = d.cu_budget > 0 ? d.cu_budget : 256;
  cus = std::max(8, std::min(cus, 256 - reserved_cus()));      // (nbdt_set_reserved_cus: a collective's CUs)
  int splits = (pp ? cus : 2 * cus) / tiles;
  const int max_splits = p.stages / 16 > 0 ? p.stages / 16 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.stages_per_split = (p.stages + splits - 1) / splits;
  splits = (p.stages + p.stages_per_split - 1) / p.stages_per_split;
  p.splits = splits;
  p.items = tiles * splits;
  p.per_xcd = (p.items + 7) / 8;
}
static int cout_tiles_wm(int cout) {
  const int mt = cout / 32;
  return mt % 5 == 0 ? 5 : (mt % 4 == 0 ? 4 : (mt % 2 == 0 ? 2 : 1));
}

// KIND 0: conv_wgrad_taps_kernel (4 waves, 2 blocks per CU), 1: conv_wgrad_pp_kernel (8 waves, two wave groups),
// 2: conv_wgrad_pp3_kernel (12 waves, three wave groups), 3: conv_wgrad_ks_kernel (8 waves, groups split the pixels).
// 1, 2 and 3 share the stage image, the split and the LDS size.
template <int WM, int KIND>
static int launch_taps(WgradTapsParams& p, hipStream_t st) {
  constexpr bool PP = KIND != 0;
  split_items(p, WM, PP);
  const size_t shmem = PP ? (size_t)5 * (64 * 64 * WM + 144 * 64) : (size_t)NSTAGE * ((32 * WM / 8) * 512 + x_bytes(4));
  const void* fn = KIND == 3 ? reinterpret_cast<const void*>(&conv_wgrad_ks_kernel<WM>)
                   : KIND == 2 ? reinterpret_cast<const void*>(&conv_wgrad_pp3_kernel<WM>)
                   : KIND == 1 ? reinterpret_cast<const void*>(&conv_wgrad_pp_kernel<WM>)
                               : reinterpret_cast<const void*>(&conv_wgrad_taps_kernel<WM, 4>);
  static DeviceAttr site;     // one per (WM, KIND) instantiation
  if (site.need(shmem)) {
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    site.done(shmem);
  }
  float* const dw = p.dw;
  const size_t dw_elems = (size_t)p.d.cout * p.d.w_ntaps * p.d.cin;
  p.dw_split_stride = 0;
  // K-split kernel: the L2 retires fp32 atomics at ~1.2 TB/s whatever their shape (profiles/r05_atomic_pattern_probe.txt),
  // so its ~47 MB of partial sums per launch are written with plain stores into one copy of dw per pixel split and a
  // streaming pass adds the copies to dw in split order (which also makes the result independent of block timing)
  // Not with more than 64 splits -- 2 tiles of a 64-channel layer -- where the fold reads more than the atomics cost, and
  // not for a CU-budgeted launch: the caller runs an HBM-bound pass beside it (the BatchNorm backward of the CU-sharing
  // schedule), the fold's ~110 MB would compete with exactly that, and the step does not get shorter
  // (profiles/r05_wgrad_store_epilogue_ab.txt: WRN-28-10 at 512 images, where every such launch is budgeted).
  // ... and only when the nine taps fill EVERY weight slot exactly once (w_ntaps == 9, w_tap a permutation): plain stores
  // into an un-zeroed copy would otherwise fold uninitialised workspace into dw (w_ntaps > 9) or lose sums (duplicate
  // w_tap).  Such descriptors keep the atomics -- into zeroed rows in deterministic mode.
  bool covers = p.d.w_ntaps == 9;
  {
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) seen |= 1u << (p.d.w_tap[t] & 31);
    covers = covers && seen == 0x1ffu;
  }
  p.store = (KIND == 3 && covers && ((wgrad_store_epilogue() && p.splits <= 64 && p.d.cu_budget == 0) || deterministic())) ? 1 : 0;
  if (p.store || deterministic()) {
    float* rows = det_rows(st, (size_t)p.splits * dw_elems);
    if (!rows && !deterministic()) {
      p.store = 0;      // no workspace (first use inside a hipGraph capture): the atomics need none
    } else {
      if (!rows) return nbdt::fail(NBDT_ENOMEM, "weight gradient: %s (%s)", "no workspace for the per-split gradients",
                                   nbdt::det_rows_why());
      if (!p.store) NBDT_HIP_CHECK(hipMemsetAsync(rows, 0, (size_t)p.splits * dw_elems * sizeof(float), st));
      p.dw = rows;
      p.dw_split_stride = (long long)dw_elems;
    }
  }
  void* args[] = {(void*)&p};
  NBDT_HIP_CHECK(hipLaunchKernel(fn, dim3(p.per_xcd * 8), dim3(KIND == 2 ? 768 : (KIND == 0 ? 256 : 512)), args, shmem, st));
  g_last_wgrad = KIND == 3 ? "conv_wgrad_ks_kernel" : KIND == 2 ? "conv_wgrad_pp3_kernel"
                           : (KIND == 1 ? "conv_wgrad_pp_kernel" : "conv_wgrad_taps_kernel");
  if (p.dw != dw) return det_fold(st, p.dw, p.splits, dw_elems, dw);
  return NBDT_OK;
}

// the 8-wave kernel: 64-pixel stages must tile the images (halo <= 144 slots) and a block needs a few dozen of
// them to amortise its prologue; the 32-bit lane offsets of its DMA need tensors below 4 GiB
static bool pp_fits_shape(const nbdt_wgrad_desc* d) {
  // (d->gw % 8: a fragment's 16-lane group holds 8 consecutive pixels of ONE image row)
  return d->gw % 8 == 0 && stage_geometry(d, 64, 144, nullptr) && (long long)d->B * d->x_bs * 2 < (1ll << 32) &&
         (long long)d->B * d->g_bs * 2 < (1ll << 32);
}
constexpr bool kKsDefault = true;        // the K-split kernel won its A/B (profiles/r05_wgrad_ksplit_ab.txt: 2-7 % by shape)
static bool takes_pp(const nbdt_wgrad_desc* d, bool pp_fits) {
  const long long M = (long long)d->B * d->gh * d->gw;
#ifndef NBDT_WPP_MIN_STAGES
#define NBDT_WPP_MIN_STAGES 64             // (round 5: 256 -> 64, profiles/r05_other_configs_wgrad_ab.txt; A/B builds: other thresholds)
#endif
  return pp_fits && (d->variant == 2 || d->variant == 4 || d->variant == 5 || (d->variant != 3 && M / 64 >= NBDT_WPP_MIN_STAGES));
}

int wgrad_taps_blocks(const nbdt_wgrad_desc* d) {
  if (!wgrad_taps_applicable(d) || !takes_pp(d, pp_fits_shape(d))) return 0;
  WgradTapsParams p;
  p.d = *d;
  stage_geometry(d, 64, 144, &p);
  split_items(p, cout_tiles_wm(d->cout), true);
  return p.per_xcd * 8 < p.items ? p.per_xcd * 8 : p.items;
}

int wgrad_taps(const nbdt_wgrad_desc* d, const void* x, const void* gy, float* dw, hipStream_t st) {
  WgradTapsParams p;
  p.d = *d;
  p.x = (const bf16_t*)x;
  p.gy = (const bf16_t*)gy;
  p.dw = dw;
  const int mt = d->cout / 32;
  const bool pp_fits = pp_fits_shape(d);
  NBDT_REQUIRE(!((d->variant == 2 || d->variant == 4 || d->variant == 5) && !pp_fits),
               "variant 2 / 4 / 5 (8- / 12-wave / K-split weight-gradient kernel): shape does not fit it");
  const bool pp = takes_pp(d, pp_fits);
  if (pp) {
    stage_geometry(d, 64, 144, &p);
    // which kernel: variant 2 / 4 / 5 force one (tap-split 8 waves / 12 waves / K-split); 0 = kKsDefault decides
    const bool ksplit = d->variant == 5 || (d->variant == 0 && kKsDefault);
    if (ksplit) {
      if (mt % 5 == 0) return launch_taps<5, 3>(p, st);
      if (mt % 4 == 0) return launch_taps<4, 3>(p, st);
      if (mt % 2 == 0) return launch_taps<2, 3>(p, st);
      return launch_taps<1, 3>(p, st);
    }
    const bool three = d->variant == 4;
    if (three) {
      if (mt % 5 == 0) return launch_taps<5, 2>(p, st);
      if (mt % 4 == 0) return launch_taps<4, 2>(p, st);
      if (mt % 2 == 0) return launch_taps<2, 2>(p, st);
      return launch_taps<1, 2>(p, st);
    }
    if (mt % 5 == 0) return launch_taps<5, 1>(p, st);
    if (mt % 4 == 0) return launch_taps<4, 1>(p, st);
    if (mt % 2 == 0) return launch_taps<2, 1>(p, st);
    return launch_taps<1, 1>(p, st);
  }
  stage_geometry(d, 32, 128, &p);
  if (mt % 5 == 0) return launch_taps<5, 0>(p, st);
  if (mt % 4 == 0) return launch_taps<4, 0>(p, st);
  if (mt % 2 == 0) return launch_taps<2, 0>(p, st);
  return launch_taps<1, 0>(p, st);
}

}  // namespace nbdt
