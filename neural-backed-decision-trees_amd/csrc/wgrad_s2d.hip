// Weight gradient of a 3x3 / stride-2 / pad-1 convolution whose input is stored as its SPACE-TO-DEPTH copy (gfx950 only).
//
// Replaces the weight-gradient half of the strided nn.Conv2d of a shape-changing unit (nbdt/models/resnet.py:56-67;
// pytorchcv PreResUnit, stride 2, behind nbdt/models/wideresnet.py:1-5).  Rounds 1-5 ran it on conv_wgrad_dma_kernel -- one
// block per (tap, cout tile, cin tile), every tap re-streaming both operands: 18.7 % MFMA-busy, 1.6x its algorithmic
// traffic, and in the training step the longest launch of its unit (350-480 us beside the data gradients).
//
// With the input as [B][H/2+2][W/2+2][4 cin] (nbdt_bn_apply_s2d) tap (r, s) of the strided conv reads phase
// (p, q) = (r != 1, s != 1) at half-resolution pixel (y - [r == 0], x - [s == 0]): every tap is a UNIT-STRIDE walk, and the
// structure of conv_wgrad_ks_kernel (wgrad_taps.hip) applies: a block owns (32 WM couts) x 32 cins x ALL NINE taps, a wave
// 16 WM couts x 16 cins x nine taps (36 WM accumulator registers); stages of 64 pixels (an rs x cs rectangle of one
// image), the two wave groups take its two 32-pixel K chunks one barrier apart; gy tile [64 px][2 CG bytes] read with
// ds_read_b64_tr_b16 once for nine taps.  What differs is the x tile: four phase sub-tiles
//     (1,1): (rs+1) x (cs+1)    (1,0): (rs+1) x cs    (0,1): rs x (cs+1)    (0,0): rs x cs      slots of 64 B (32 cins)
// -- (2rs+1)(2cs+1) slots, 289-297 for the 8x8 / 16x16 grids of WRN-28-10 against the dense kernel's 100-108: the strided
// conv reads 2.75x the pixels per MFMA -- and per kernel row a wave reads a 9-pixel run of the q = 1 sub-tile (taps s = 0 and
// s = 2, three transpose reads) and an 8-pixel run of the q = 0 one (tap s = 1, two): 15 + 2 WM reads per 9 WM MFMAs (the
// dense kernel: 9 + 2 WM).  Ring of 4 stages (40 KB each at WM = 5), prefetch distance 3, counted vmcnt as in the parent.
#include "common.h"
#include <stdlib.h>
#include <algorithm>

using namespace nbdt;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

namespace nbdt {
struct WgradS2dParams {
  nbdt_wgrad_desc d;
  const bf16_t* x;
  const bf16_t* gy;
  float* dw;
  long long dw_split_stride;
  int stages, stages_per_split, n_ci_blocks, splits, items, per_xcd;
  int store;
  int rs, cs;            // stage rectangle, rs * cs == 64
  int x_slots, x_instr;  // (2 rs + 1)(2 cs + 1) slots, ceil(x_slots / 16) LDS-DMA pieces
  int stages_per_row, rowgroups;
  int x_pix;             // elements per pixel of the space-to-depth tensor (4 cin)
  FastDiv div_spr, div_rg;
};
}  // namespace nbdt

template <int WM>
__global__ __launch_bounds__(512, 2) void conv_wgrad_s2d_kernel(nbdt::WgradS2dParams p) {
  constexpr int CG = 32 * WM;
  constexpr int KSP = 64;
  constexpr int PG = 2 * CG;
  constexpr int G_BYTES = KSP * PG;
  constexpr int G_INSTR = G_BYTES / 1024;
  constexpr int IPG = (G_INSTR + 7) / 8;
  constexpr int IPX = 3;              // x pieces per wave (x_instr <= 24)
  constexpr int NSLOT = 4, PD = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const nbdt_wgrad_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
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
  const int rs = NBDT_PIN(p.rs), cs = NBDT_PIN(p.cs);
  const int x_instr = NBDT_PIN(p.x_instr), x_slots = NBDT_PIN(p.x_slots);
  const int x_bytes = x_instr * 1024;
  const int stage_bytes = G_BYTES + x_bytes;
  FastDiv dspr, drg;
  dspr.mul = NBDT_PIN(p.div_spr.mul); dspr.sh = NBDT_PIN(p.div_spr.sh); dspr.d = NBDT_PIN(p.div_spr.d);
  drg.mul = NBDT_PIN(p.div_rg.mul); drg.sh = NBDT_PIN(p.div_rg.sh); drg.d = NBDT_PIN(p.div_rg.d);
  const unsigned long long gy_u = (unsigned long long)p.gy, x_u = (unsigned long long)p.x;
  const bf16_t* gy_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(gy_u >> 32)) << 32) |
                                          (unsigned)NBDT_PIN((unsigned)gy_u));
  const bf16_t* x_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(x_u >> 32)) << 32) |
                                         (unsigned)NBDT_PIN((unsigned)x_u));
#undef NBDT_PIN
  // sub-tile (p, q): rows rs + p, columns cs + q, first slot sub_off; slot (i, j) is half-resolution pixel
  // (row0 - p + i, col0 - q + j) of the stage's image = padded pixel (+1, +1), channels (2p + q) cin + ci0 ..
  const int C1 = cs + 1, R1 = rs + 1;
  const int off11 = 0, off10 = R1 * C1, off01 = off10 + R1 * cs, off00 = off01 + rs * C1;

  // ---- LDS stage image, DMA pieces.  gy: conv_wgrad_ks_kernel's.  x: slot = 16 id + (lane >> 2), 16-byte chunk
  // position lane & 3 holding source chunk (lane & 3) ^ (2 * (((j >> 3) ^ i) & 1)): the two 32-byte halves (16 cins each)
  // of a slot swap with the parity of (j >> 3) ^ i, like the parent's halo tile.
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
  unsigned x_voff[IPX];
#pragma unroll
  for (int k = 0; k < IPX; ++k) {
    int slot = 16 * (wave + 8 * k) + (lane >> 2);
    slot = slot < x_slots ? slot : x_slots - 1;
    int pp, qq, rel;
    if (slot < off10) { pp = 1; qq = 1; rel = slot - off11; }
    else if (slot < off01) { pp = 1; qq = 0; rel = slot - off10; }
    else if (slot < off00) { pp = 0; qq = 1; rel = slot - off01; }
    else { pp = 0; qq = 0; rel = slot - off00; }
    const int cq = cs + qq;
    const int i = rel / cq, j = rel - i * cq;
    const int src_chunk = (lane & 3) ^ ((((j >> 3) ^ i) & 1) << 1);
    x_voff[k] = (unsigned)((i - pp + 1) * x_hs + (j - qq + 1) * x_ws + (2 * pp + qq) * d.cin + ci0 + src_chunk * 8) * 2u;
  }
  int n_mine = 0;
#pragma unroll
  for (int k = 0; k < IPX; ++k) n_mine += (wave + 8 * k < x_instr) ? 1 : 0;
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
    const unsigned dst0 = lds_base + slot_i * stage_bytes;
#pragma unroll
    for (int k = 0; k < IPG; ++k)
      if (wave + 8 * k < G_INSTR) glds16_s(gy_base + g_stage, g_voff[k], dst0 + (wave + 8 * k) * 1024);
#pragma unroll
    for (int k = 0; k < IPX; ++k)
      if (wave + 8 * k < x_instr) glds16_s(x_base + x_stage, x_voff[k], dst0 + G_BYTES + (wave + 8 * k) * 1024);
  };

  f32x4 acc[9][WM];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < WM; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- transpose-read addressing: this group's K chunk = pixels 32 grp .. 32 grp + 31 of every stage; a 16-lane group
  // (g4) holds 8 consecutive pixels of one image row, lane t16 reads pixel row pr = t16 >> 2 of a 4-pixel block
  typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;
  typedef __attribute__((address_space(3))) s16x4* lds_tr;
  const lds_cptr smem3 = (lds_cptr)smem;
  const int g4 = lane >> 4, t16 = lane & 15;
  const int pr = t16 >> 2;
  const int c8 = (t16 & 3) * 8;
  int g_lane_off[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
    g_lane_off[a] = (grp * 32 + 8 * g4 + pr) * PG + (((wm * WM + a) ^ (g4 & 1)) * 32) + c8;
  // Per kernel row: the 9-pixel run of the q = 1 sub-tile is blocks 0, 1 (one address + 256 B: columns j .. j + 7 share
  // j >> 3, i.e. the half swap, because col_l is a multiple of 8) and block 2 (its own address: j >> 3 is one more); the
  // 8-pixel run of the q = 0 sub-tile is one address + 256 B.  Nine address registers, as in the parent.
  int xrel1[3], xrel1c[3], xrel0[3];
  {
    const int k = grp * 32 + 8 * g4;
    const int row_l = k / cs, col_l = k - row_l * cs;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int pp = r != 1 ? 1 : 0;
      const int i = row_l + (r == 2 ? 1 : 0);           // r == 0: row0 - 1 + row_l is slot row row_l of a p = 1 sub-tile
      const int sub1 = pp ? off11 : off01, sub0 = pp ? off10 : off00;
      const int j = col_l + pr, jc = j + 8;
      xrel1[r] = G_BYTES + c8 + (((sub1 + i * C1 + j) << 6) | (((((j >> 3) ^ i) ^ wn) & 1) << 5));
      xrel1c[r] = G_BYTES + c8 + (((sub1 + i * C1 + jc) << 6) | (((((jc >> 3) ^ i) ^ wn) & 1) << 5));
      xrel0[r] = G_BYTES + c8 + (((sub0 + i * cs + j) << 6) | (((((j >> 3) ^ i) ^ wn) & 1) << 5));
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

  // fragment addresses of ring slot 0; after every stage they advance by one slot (or wrap): one scalar delta added to each
  // (keeping the slot-relative offsets in registers as well cost the 9 + 2 registers this kernel does not have)
  lds_cptr xa1[3], xa1c[3], xa0[3];
  lds_cptr ga[2];
  ga[0] = smem3 + g_lane_off[0];
  ga[1] = smem3 + g_lane_off[1];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    xa1[r] = smem3 + xrel1[r];
    xa1c[r] = smem3 + xrel1c[r];
    xa0[r] = smem3 + xrel0[r];
  }
  auto advance = [&](int from_slot, int to_slot) {
    int delta = (to_slot - from_slot) * stage_bytes;
    asm volatile("" : "+s"(delta));
    ga[0] += delta;
    ga[1] += delta;
#pragma unroll
    for (int r = 0; r < 3; ++r) { xa1[r] += delta; xa1c[r] += delta; xa0[r] += delta; }
  };
  int slot_c = 0, slot_n = 1 % NSLOT, slot_d = PD % NSLOT;
  int g_next = 0, x_next = 0;
  if (PD < n_st) stage_off(s_begin + PD, g_next, x_next);

  for (int u = 0; u < n_st; ++u) {
    // ================= L(u): this wave's fragments of its chunk -> registers, its pieces of stage u + PD =================
    bf16x8 gf[WM];
    u32x2 xw1[3][3], xw0[3][2];
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const lds_cptr a0 = ga[a & 1] + (a >> 1) * 64;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(a0 + 4 * PG));
      gf[a] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      xw1[r][0] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)xa1[r]));
      xw1[r][1] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(xa1[r] + 256)));
      xw1[r][2] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)xa1c[r]));
      xw0[r][0] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)xa0[r]));
      xw0[r][1] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(xa0[r] + 256)));
    }
    // stage u+1 (issued two load segments ago) must be in LDS before the next barrier; the stage issued since may fly
    if (u + PD <= n_st) {
      switch (n_mine) {
#define NBDT_CASE(K) case K: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
        NBDT_CASE(1) NBDT_CASE(2) NBDT_CASE(3) NBDT_CASE(4) NBDT_CASE(5) NBDT_CASE(6)
#undef NBDT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (u + PD < n_st) issue(slot_d, g_next, x_next);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ================= M(u): 9 x WM MFMAs; their shadow prepares L(u+1) =================
    __builtin_amdgcn_s_setprio(1);
    advance(slot_c, slot_n);
    stage_off(s_begin + u + 1 + PD, g_next, x_next);       // (past the end: computed, never used)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int r = t / 3, sft = t % 3;
      u32x4_t v;
      if (sft == 1) {                   // q = 0 sub-tile, pixels col_l .. col_l + 7
        v = u32x4_t{xw0[r][0][0], xw0[r][0][1], xw0[r][1][0], xw0[r][1][1]};
      } else {
        const u32x2 A = xw1[r][0], B = xw1[r][1], C = xw1[r][2];
        if (sft == 0) v = u32x4_t{A[0], A[1], B[0], B[1]};      // q = 1 sub-tile column j is image column col0 - 1 + j
        else v = u32x4_t{__builtin_amdgcn_alignbit(A[1], A[0], 16), __builtin_amdgcn_alignbit(B[0], A[1], 16),
                         __builtin_amdgcn_alignbit(B[1], B[0], 16), __builtin_amdgcn_alignbit(C[0], B[1], 16)};
      }
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
    for (int r = 0; r < 3; ++r) asm volatile("" : "+v"(xa1[r]), "+v"(xa1c[r]), "+v"(xa0[r]));
    asm volatile("" : "+v"(ga[0]), "+v"(ga[1]));
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    slot_c = slot_n;
    slot_n = slot_n + 1 == NSLOT ? 0 : slot_n + 1;
    slot_d = slot_d + 1 == NSLOT ? 0 : slot_d + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  // ---- the two groups' partial sums meet through LDS (lane to lane with the partner wave w ^ 4, two rounds; the ring is
  // dead: every wave is past its last read and no LDS-DMA is in flight), then taps 0-4 are group 0's, taps 5-8 group 1's
  __builtin_amdgcn_sched_barrier(0);
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
  const int ci = ci0 + wn * 16 + t16;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if ((t < 5) != (grp == 0)) continue;       // wave-uniform
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

namespace nbdt {

static int s2d_wm(int cout) {
  const int mt = cout / 32;
  return mt % 5 == 0 ? 5 : (mt % 4 == 0 ? 4 : (mt % 2 == 0 ? 2 : 1));
}

// The descriptor ops.conv_wgrad_desc_s2d builds (and nothing else): nine taps over a tensor of 4 cin-element pixels,
// tap (r, s) at ((r != 0) * x_hs + (s != 0) * x_ws + (2 [r != 1] + [s != 1]) cin) from the tensor's first element.
static bool s2d_geometry(const nbdt_wgrad_desc* d, WgradS2dParams* p) {
  const int gw = d->gw, gh = d->gh, ks = 64;
  if (gw % 8 != 0 || !(gw % ks == 0 || ks % gw == 0)) return false;
  int rs = gw >= ks ? 1 : ks / gw;
  int cs = gw >= ks ? ks : gw;
  if (gw >= 64) { if (gw % 32 != 0 || gh % 2 != 0) return false; rs = 2; cs = 32; }
  if (gh % rs != 0) return false;
  const int slots = (2 * rs + 1) * (2 * cs + 1);
  const int x_instr = (slots + 15) / 16;
  if (x_instr > 24) return false;
  const size_t stage = (size_t)64 * 64 * s2d_wm(d->cout) + (size_t)x_instr * 1024;
  if (4 * stage > 160 * 1024) return false;
  if (!p) return true;
  p->rs = rs; p->cs = cs;
  p->x_slots = slots; p->x_instr = x_instr;
  p->stages = (int)((long long)d->B * gh * gw / ks);
  p->stages_per_row = gw / cs;
  p->rowgroups = gh / rs;
  p->div_spr = make_fastdiv((unsigned)p->stages_per_row);
  p->div_rg = make_fastdiv((unsigned)p->rowgroups);
  return true;
}

bool wgrad_s2d_applicable(const nbdt_wgrad_desc* d) {
  if (d->ntaps != 9 || d->w_ntaps != 9 || d->x_base != 0) return false;
  if (d->x_ws != 4 * d->cin || d->x_hs != (d->gw + 2) * d->x_ws || d->x_bs != (d->gh + 2) * d->x_hs) return false;
  if (d->g_ws != d->cout || d->g_hs != (d->gw + 2) * d->cout) return false;
  for (int t = 0; t < 9; ++t) {
    const int r = t / 3, s = t % 3;
    if (d->w_tap[t] != t) return false;
    if (d->tap_off[t] != (r != 0) * d->x_hs + (s != 0) * d->x_ws + (2 * (r != 1) + (s != 1)) * d->cin) return false;
  }
  if ((long long)d->B * d->x_bs * 2 >= (1ll << 32) || (long long)d->B * d->g_bs * 2 >= (1ll << 32)) return false;
  if ((long long)d->B * d->gh * d->gw < 64 * 64) return false;       // (a few dozen stages per block or the old kernel)
  return s2d_geometry(d, nullptr);
}

// pixel split of a launch: one block per CU, whole rounds filled from below (d.cu_budget: the caller wants only that many
// CUs filled -- an HBM-bound pass on another stream gets the rest), at least 16 stages per block
static void s2d_split(WgradS2dParams& p, int wm) {
  const nbdt_wgrad_desc& d = p.d;
  p.n_ci_blocks = d.cin / 32;
  const int tiles = (d.cout / (32 * wm)) * p.n_ci_blocks;
  int cus = d.cu_budget > 0 ? d.cu_budget : 256;
  cus = std::max(8, std::min(cus, 256 - reserved_cus()));
  int splits = cus / tiles;
  const int max_splits = p.stages / 16 > 0 ? p.stages / 16 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.stages_per_split = (p.stages + splits - 1) / splits;
  splits = (p.stages + p.stages_per_split - 1) / p.stages_per_split;
  p.splits = splits;
  p.items = tiles * splits;
  p.per_xcd = (p.items + 7) / 8;
}

int wgrad_s2d_blocks(const nbdt_wgrad_desc* d) {
  WgradS2dParams p;
  p.d = *d;
  if (!s2d_geometry(d, &p)) return 0;
  s2d_split(p, s2d_wm(d->cout));
  return p.per_xcd * 8 < p.items ? p.per_xcd * 8 : p.items;
}

template <int WM>
static int launch_s2d(WgradS2dParams& p, hipStream_t st) {
  const nbdt_wgrad_desc& d = p.d;
  s2d_split(p, WM);
  const size_t shmem = (size_t)4 * (64 * 64 * WM + (size_t)p.x_instr * 1024);
  const void* fn = reinterpret_cast<const void*>(&conv_wgrad_s2d_kernel<WM>);
  static DeviceAttr site;
  if (site.need(shmem)) {
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    site.done(shmem);
  }
  float* const dw = p.dw;
  const size_t dw_elems = (size_t)d.cout * d.w_ntaps * d.cin;
  p.dw_split_stride = 0;
  p.store = ((wgrad_store_epilogue() && p.splits <= 64 && d.cu_budget == 0) || deterministic()) ? 1 : 0;
  if (p.store) {
    float* rows = det_rows(st, (size_t)p.splits * dw_elems);
    if (!rows && !deterministic()) {
      p.store = 0;
    } else {
      if (!rows) return nbdt::fail(NBDT_ENOMEM, "weight gradient: %s (%s)", "no workspace for the per-split gradients",
                                   nbdt::det_rows_why());
      p.dw = rows;
      p.dw_split_stride = (long long)dw_elems;
    }
  }
  void* args[] = {(void*)&p};
  NBDT_HIP_CHECK(hipLaunchKernel(fn, dim3(p.per_xcd * 8), dim3(512), args, shmem, st));
  g_last_wgrad = "conv_wgrad_s2d_kernel";
  if (p.dw != dw) return det_fold(st, p.dw, p.splits, dw_elems, dw);
  return NBDT_OK;
}

int wgrad_s2d(const nbdt_wgrad_desc* d, const void* x, const void* gy, float* dw, hipStream_t st) {
  WgradS2dParams p;
  p.d = *d;
  p.x = (const bf16_t*)x;
  p.gy = (const bf16_t*)gy;
  p.dw = dw;
  p.x_pix = 4 * d->cin;
  if (!s2d_geometry(d, &p)) return nbdt::fail(NBDT_EINVAL, "%s%s", "space-to-depth weight gradient: shape does not fit", "");
  const int wm = s2d_wm(d->cout);
  if (wm == 5) return launch_s2d<5>(p, st);
  if (wm == 4) return launch_s2d<4>(p, st);
  if (wm == 2) return launch_s2d<2>(p, st);
  return launch_s2d<1>(p, st);
}

}  // namespace nbdt
