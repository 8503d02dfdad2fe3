// 3x3 / stride-1 / pad-1 implicit GEMM with an LDS-resident HALO tile (gfx950 only).
//
// Ablation of conv_igemm_dma_kernel on MI355X (B=512, 32x32, 160->160): compute-only loop 196 us,
// data-movement-only 212 us, both 294 us -- the kernel pulls 2.4 GB per launch through L2 (~15 TB/s)
// because every one of the 9 taps re-loads the (shifted) 256-pixel input tile.  Here a block loads,
// per 32-channel slice, the input rows it needs INCLUDING the one-pixel halo ONCE
// ((RB+2) x (W+2) pixels x 64 B: 21.8 KB instead of 9 x 16 KB) and the 9 taps read shifted windows of
// that tile straight from LDS: tap (r,s) is a constant offset of r*(W+2)+s halo pixels for every lane.
// The halo of neighbouring tiles / images is the zero border of the padded NHWC layout, so there is still
// no predication.  Weight tiles stream through a 3-slot ring exactly as in conv_dma.hip.
//
//   K steps t = (kc, tap), tap fastest.  Per step: wait(vmcnt) -> s_barrier -> issue DMA -> MFMAs.
//     A halo tile of slice kc+1 -> other A buffer, issued at (kc, tap 0) before W(t+2)
//     W(t+2) -> ring slot (t+2)%3
//   In-order completion of DMA loads makes "W(t) landed" imply "A(kc) landed" (issued >= 9 steps earlier).
//
// Same block tile (256 px x 32*NT couts), MFMA mapping, LDS swizzle (by halo-pixel index) and epilogue
// as the generic kernel.  Used for forward and stride-1 data-gradient 3x3 convs whose pixel tile is whole
// image rows / whole images (32x32, 16x16, 8x8, 64x64, ...): 97% of the backbone's igemm flops.
#include "conv_common.h"

#ifndef NBDT_HALO_NWS8
#define NBDT_HALO_NWS8 3    // weight-ring slots of the 8-wave kernel (prefetch distance = slots - 1)
#endif
__device__ __forceinline__ void gload16(u32x4_t& dst, const void* p) {   // asynchronous: see the counted waits
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void touch16(u32x4_t& v) { asm volatile("" : "+v"(v)::"memory"); }

#ifndef NBDT_HALO_WREG
#define NBDT_HALO_WREG 0    // 1: weight tiles go global_load -> VGPR -> ds_write (only the halo uses LDS-DMA);
                            // measured 3-7 % SLOWER per launch than the all-DMA ring on MI355X -> off
#endif
constexpr int nw_slots(int nwv) { return NBDT_HALO_WREG ? 2 : (nwv == 8 ? NBDT_HALO_NWS8 : 3); }
#ifndef NBDT_HALO_DEBUG
#define NBDT_HALO_DEBUG 0   // 1: no DMA, 2: no waits/barriers, 4: no MFMA, 8: no LDS fragment reads,
                            // 16: every DMA reads the same 1 KB (timing only)
#endif

constexpr int min_w_dma_h(int w_instr, int nwv) {
  int best = 1 << 30;
  for (int w = 0; w < nwv; ++w) {
    int n = 0;
    for (int id = w; id < w_instr; id += nwv) ++n;
    best = n < best ? n : best;
  }
  return best;
}


// NWV = waves per block: 4 (256-pixel tile, 2 blocks per CU) or 8 (512-pixel tile, 1 block per CU).
// Ablation of the 4-wave kernel (compile-time NBDT_HALO_DEBUG variants, 160->160 @ 32x32, same box):
// MFMA only 125 us, + LDS fragment reads 155, DMA + barriers only 139, everything 280 -- the DMA stream
// (1.14 GB per launch through L2, 80 % of it the weight tile every 256-pixel block re-fetches per tap) takes
// as long as the math and does not hide behind it: a wave blocked issuing DMA cannot issue MFMAs.  A
// 512-pixel tile halves the weight traffic per MFMA and the DMA instructions per wave.
template <int NT, bool HAS_RES, int STATS, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void conv3x3_halo_kernel(nbdt::ConvDmaParams p,
                                                                                   nbdt::HaloGeom hg) {
  constexpr int BN = 32 * NT;
  constexpr int BMH = 64 * NWV;                     // pixels per block
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int W_INSTR = W_BYTES / 1024;
  constexpr int IPW_W = (W_INSTR + NWV - 1) / NWV;
  constexpr int MINW = min_w_dma_h(W_INSTR, NWV);
  constexpr int MAX_A_SLOTS = NWV == 4 ? 10 : 7;
  constexpr int NWS = nw_slots(NWV);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [A buf 0][A buf 1][W ring x3]

  const int bid = blockIdx.x;
  const int item = (bid & 7) * p.per_xcd + (bid >> 3);
  if (item >= p.m_blocks * p.n_blocks) return;
  const int m_blk = item / p.n_blocks;
  const int n_blk = item - m_blk * p.n_blocks;
  const int m0 = m_blk * BMH;
  const int n0 = n_blk * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const nbdt_conv_desc& d = p.d;

#define NBDT_PIN(x) __builtin_amdgcn_readfirstlane(x)
  const int cin = NBDT_PIN(d.cin);
  const int kchunks = cin >> 5;
  const int nk = 9 * kchunks;
  const int w_row_len = d.w_ntaps * cin;
  const int a_bytes = NBDT_PIN(hg.a_bytes);
  const int a_instr = NBDT_PIN(hg.a_instr);
  const int a_slots = (a_instr - wave + NWV - 1) / NWV;   // DMA instructions this wave issues per halo tile
  const int a_min = a_instr / NWV;                        // fewest any wave issues (for the counted waits)
  const int hw2 = NBDT_PIN(hg.hw2), himg = NBDT_PIN(hg.himg), hp_total = NBDT_PIN(hg.hp);
  const bool tiled = __builtin_amdgcn_readfirstlane(p.w_tiled != nullptr ? 1 : 0) != 0;
  const unsigned long long in_u = (unsigned long long)p.in, w_u = (unsigned long long)(tiled ? p.w_tiled : p.w);
  const bf16_t* in_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(in_u >> 32)) << 32) |
                                          (unsigned)NBDT_PIN((unsigned)in_u));
  const bf16_t* w_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(w_u >> 32)) << 32) |
                                         (unsigned)NBDT_PIN((unsigned)w_u));
#undef NBDT_PIN
  const int tap_w_v = lane < 9 ? d.w_tap[lane < 9 ? lane : 0] * cin : 0;   // lane t holds tap t's weight k-offset

  // ---- which images / rows this pixel tile covers
  const int img_px = d.gh * d.gw;
  int b0, row0;
  if (hg.ib == 1) {
    b0 = m_blk / hg.blocks_per_img;
    row0 = (m_blk - b0 * hg.blocks_per_img) * hg.rb;
  } else {
    b0 = m_blk * hg.ib;
    row0 = 0;
  }
  (void)img_px;

  // ---- A halo DMA slots: wave-instruction id = wave + NWV*k covers halo pixels [16*id, 16*id+16)
  const int cpos = lane & 3;
  int a_src[MAX_A_SLOTS];
#pragma unroll
  for (int k = 0; k < MAX_A_SLOTS; ++k) {
    int hp = (wave + NWV * k) * 16 + (lane >> 2);
    const int swz = (hp >> 2) & 3;            // swizzle follows the LDS position, not the clamped pixel
    hp = hp < hp_total ? hp : hp_total - 1;   // tail lanes re-fetch the last halo pixel (harmless)
    const int img = hp / himg;
    const int rem = hp - img * himg;
    const int hr = rem / hw2, hc = rem - hr * hw2;
    int b = b0 + img;
    b = b < d.B ? b : d.B - 1;                // M tail: whole images past the batch re-read the last one
    a_src[k] = b * d.in_bs + (row0 + hr) * d.in_hs + hc * d.in_ws + ((cpos ^ swz) << 3);
  }
  int w_src[IPW_W];
#pragma unroll
  for (int k = 0; k < IPW_W; ++k) {
    const int id = wave + NWV * k;
    int row = id * 16 + (lane >> 2);
    row = row < BN ? row : BN - 1;
    // tiled weights: the tile IS the LDS image -> instruction id reads its own contiguous KiB
    w_src[k] = tiled ? n_blk * kchunks * 9 * (BN * 32) + id * 512 + lane * 8
                     : (n0 + row) * w_row_len + ((cpos ^ ((row >> 2) & 3)) << 3);
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned w_ring = lds_base + 2 * a_bytes;

  constexpr int dbg = NBDT_HALO_DEBUG;   // compile-time timing experiments only (scratch/ablate.sh)
  auto issue_a = [&](int buf, int kc) {
    const unsigned dst0 = lds_base + buf * a_bytes;
#pragma unroll
    for (int k = 0; k < MAX_A_SLOTS; ++k)
      if (k < a_slots)   // wave-uniform
        glds16(in_base + ((dbg & 16) ? (lane << 3) : (a_src[k] + kc * BK)),
               __builtin_amdgcn_readfirstlane(dst0 + (wave + NWV * k) * 1024));
  };
  auto issue_w = [&](int slot, int tap, int kc) {
    const int w_k = tiled ? (kc * 9 + tap) * (BN * 32) : __builtin_amdgcn_readlane(tap_w_v, tap) + kc * BK;
    const unsigned dst0 = w_ring + slot * W_BYTES;
#pragma unroll
    for (int k = 0; k < IPW_W; ++k) {
      const int id = wave + NWV * k;
      if (id < W_INSTR)
        glds16(w_base + ((dbg & 16) ? (lane << 3) : (w_src[k] + w_k)), __builtin_amdgcn_readfirstlane(dst0 + id * 1024));
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
  // halo index of this lane's two output pixels at tap (0,0)
  int hp0[2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int pl = wave * 64 + tm * 32 + frag_row;     // pixel inside the block's pixel tile
    const int per_img = hg.rb * d.gw;
    const int img = pl / per_img;
    const int rem = pl - img * per_img;
    const int r = rem / d.gw, c = rem - r * d.gw;
    hp0[tm] = img * himg + r * hw2 + c;
  }

  // (Tried on MI355X and dropped, same box A/B: hoisting all 14 ds_read_b128 of a tap above the DMA issue,
  //  and a two-register-set software pipeline with the barrier between the two 16-channel halves of a tap --
  //  neither moved the MFMA+LDS-only time of 165 us: the gap to the MFMA-only 134 us is operand data, not
  //  LDS latency; the MFMA-only variant multiplies zeros and clocks higher.)
  auto compute = [&](int abuf, int wslot, int tap) {
    if (dbg & 8) {   // MFMA-only: no LDS reads
      bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      asm volatile("" : "+v"(z));
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z, z, acc[tn][tm], 0, 0, 0);
      return;
    }
    const unsigned char* As = smem + abuf * a_bytes;
    const unsigned char* Ws = smem + 2 * a_bytes + wslot * W_BYTES;
    const int tr = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0);
    const int toff = tr * hw2 + (tap - 3 * tr);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 2 * ks + frag_half;
      bf16x8 pf[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) pf[tm] = *(const bf16x8*)(As + lds_off(hp0[tm] + toff, c));
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const bf16x8 wf = *(const bf16x8*)(Ws + lds_off(tn * 32 + frag_row, c));
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, pf[tm], acc[tn][tm], 0, 0, 0);
      }
    }
  };

  // counted wait: `n` most recent DMA instructions of this wave may stay in flight (immediate operand)
  auto wait_vm = [&](int n) {
    switch (n) {
#define NBDT_CASE(K) case K: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K) : "memory"); break;
      NBDT_CASE(1) NBDT_CASE(2) NBDT_CASE(3) NBDT_CASE(4) NBDT_CASE(5) NBDT_CASE(6) NBDT_CASE(7) NBDT_CASE(8)
      NBDT_CASE(9) NBDT_CASE(10) NBDT_CASE(11) NBDT_CASE(12) NBDT_CASE(13) NBDT_CASE(14) NBDT_CASE(15)
      NBDT_CASE(16) NBDT_CASE(17) NBDT_CASE(18)
#undef NBDT_CASE
      default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    }
  };

  if constexpr (NBDT_HALO_WREG) {
    // ---- weights through registers.  Ablation: with the LDS-DMA instructions merely ISSUED (never waited
    // for) the loop slows from 165 to 230-250 us -- a wave stuck issuing DMA cannot issue MFMAs -- and 70 % of
    // the DMA instructions are weight tiles.  Here W(u+2) is fetched with plain global_load_dwordx4 at step u
    // (two register sets, two steps of latency budget), written to the 2-slot ring with ds_write_b128 at the
    // top of step u+2 just before that step's barrier; only the halo tile (once per 9 steps) still uses DMA.
    //   issue order:  ... Wl(u) | A?(u-2) | Wl(u+1) | A?(u-1) | <top of step u>
    int cnt_w = 0;
#pragma unroll
    for (int k = 0; k < IPW_W; ++k) cnt_w += (wave + NWV * k < W_INSTR) ? 1 : 0;
    u32x4_t wreg[2][IPW_W];
    auto load_w = [&](int set, int tap_, int kc_) {
      const int w_k = tiled ? (kc_ * 9 + tap_) * (BN * 32) : __builtin_amdgcn_readlane(tap_w_v, tap_) + kc_ * BK;
#pragma unroll
      for (int k = 0; k < IPW_W; ++k)
        if (wave + NWV * k < W_INSTR) gload16(wreg[set][k], w_base + (w_src[k] + w_k));
    };
    auto commit_w = [&](int set, int slot) {
#pragma unroll
      for (int k = 0; k < IPW_W; ++k) {   // the wait above is only meaningful if the values are taken from here on
        touch16(wreg[set][k]);
        const int id = wave + NWV * k;
        if (id < W_INSTR) *(u32x4_t*)(smem + 2 * a_bytes + slot * W_BYTES + id * 1024 + lane * 16) = wreg[set][k];
      }
    };
    auto wait_n = [&](int n) {
      switch (n) {
#define NBDT_CASE(K) case K: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
        NBDT_CASE(1) NBDT_CASE(2) NBDT_CASE(3) NBDT_CASE(4) NBDT_CASE(5) NBDT_CASE(6) NBDT_CASE(7) NBDT_CASE(8)
        NBDT_CASE(9) NBDT_CASE(10) NBDT_CASE(11) NBDT_CASE(12) NBDT_CASE(13) NBDT_CASE(14) NBDT_CASE(15)
        NBDT_CASE(16) NBDT_CASE(17) NBDT_CASE(18) NBDT_CASE(19) NBDT_CASE(20) NBDT_CASE(21) NBDT_CASE(22)
        NBDT_CASE(23) NBDT_CASE(24) NBDT_CASE(25) NBDT_CASE(26)
#undef NBDT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    };
    if (!(dbg & 1)) {
      issue_a(0, 0);
      load_w(0, 0, 0);
      load_w(1, 1, 0);
    }
    int tap = 0, kc = 0;
    int a_prev1 = 0, a_prev2 = 0;      // DMA instructions this wave issued for a halo tile at steps u-1 / u-2
    auto step = [&](int SET, int u) {   // SET is a literal at both call sites: register sets resolve statically
      if (!(dbg & 2)) wait_n(a_prev2 + (u + 1 < nk ? cnt_w : 0) + a_prev1);
      if (!(dbg & 1)) commit_w(SET, SET);
      if (!(dbg & 2)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::: "memory");
      a_prev2 = a_prev1;
      a_prev1 = 0;
      if (!(dbg & 1)) {
        if (u + 2 < nk) {
          int t2 = tap + 2, k2 = kc;
          if (t2 >= 9) { t2 -= 9; ++k2; }
          load_w(SET, t2, k2);
        }
        if (tap == 0 && kc + 1 < kchunks) {
          issue_a((kc + 1) & 1, kc + 1);
          a_prev1 = a_slots;
        }
      }
      if (!(dbg & 4)) compute(kc & 1, SET, tap);
      if (++tap == 9) { tap = 0; ++kc; }
    };
    for (int u = 0; u < nk; u += 2) {
      step(0, u);
      if (u + 1 < nk) step(1, u + 1);
    }
  } else {
  // ---- pipeline: W tiles are prefetched PD = NWS-1 steps ahead through an NWS-slot ring
  constexpr int PD = NWS - 1;
  if (!(dbg & 1)) {
    issue_a(0, 0);
#pragma unroll
    for (int i = 0; i < PD; ++i) issue_w(i, i, 0);     // nk >= 9 > PD: taps 0..PD-1 of slice 0
  }
  int tap = 0, kc = 0, wslot = 0;
  int a_age = 1 << 20;   // steps since the last A halo tile was issued (it sits between two W tiles in issue order)
  for (int t = 0; t < nk; ++t) {
    if (!(dbg & 2)) {
      // issued after W(t): W(t+1) .. W(min(t+PD-1, nk-1)), plus an A tile if one was issued in the last PD-1 steps
      int after = nk - 1 - t;
      after = after < PD - 1 ? after : PD - 1;
      int n = after * MINW;
      if (a_age <= PD - 1 && after > 0) n += a_min;
      wait_vm(n);
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("" ::: "memory");
    ++a_age;
    if (tap == 0 && kc + 1 < kchunks && !(dbg & 1)) {
      issue_a((kc + 1) & 1, kc + 1);
      a_age = 1;
    }
    if (t + PD < nk && !(dbg & 1)) {
      int t2 = tap + PD, k2 = kc;
      if (t2 >= 9) { t2 -= 9; ++k2; }
      int s2 = wslot + PD;
      s2 = s2 >= NWS ? s2 - NWS : s2;
      issue_w(s2, t2, k2);
    }
    if (!(dbg & 4)) compute(kc & 1, wslot, tap);
    wslot = wslot + 1 == NWS ? 0 : wslot + 1;
    if (++tap == 9) { tap = 0; ++kc; }
  }
  }  // DMA weight path

  conv_epilogue<NT, HAS_RES, STATS, NWV>(acc, p, smem, m0, n0, m_blk, wave, lane, tid);
}

namespace nbdt {

template <int NT, int NWV>
static int launch_halo(ConvDmaParams& p, const HaloGeom& hg, hipStream_t st) {
  constexpr int BN = 32 * NT;
  constexpr int BMH = 64 * NWV;
  p.n_blocks = p.d.cout / BN;
  p.m_blocks = (p.M + BMH - 1) / BMH;
  const int items = p.m_blocks * p.n_blocks;
  p.per_xcd = (items + 7) / 8;
  size_t shmem = 2 * (size_t)hg.a_bytes + (size_t)nw_slots(NWV) * BN * BK * 2;
  const size_t epi = conv_epilogue_lds_bytes<NT, NWV>();
  if (shmem < epi) shmem = epi;
  static size_t attr_bytes = 0;
  if (shmem > attr_bytes) {
#define NBDT_ATTR(R, S)                                                                                     \
  NBDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<NT, R, S, NWV>),    \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))
    NBDT_ATTR(true, 1); NBDT_ATTR(true, 0); NBDT_ATTR(false, 1); NBDT_ATTR(false, 0); NBDT_ATTR(false, 2);
    NBDT_ATTR(true, 3); NBDT_ATTR(false, 3);
#undef NBDT_ATTR
    attr_bytes = shmem;
  }
  const dim3 grid(p.per_xcd * 8), blk(64 * NWV);
#define NBDT_GO(R, S) hipLaunchKernelGGL((conv3x3_halo_kernel<NT, R, S, NWV>), grid, blk, shmem, st, p, hg)
  if (p.aff_scale != nullptr) { if (p.res != nullptr) NBDT_GO(true, 3); else NBDT_GO(false, 3); }
  else if (p.bn_x != nullptr) NBDT_GO(false, 2);
  else if (p.res != nullptr) { if (p.stats) NBDT_GO(true, 1); else NBDT_GO(true, 0); }
  else { if (p.stats) NBDT_GO(false, 1); else NBDT_GO(false, 0); }
#undef NBDT_GO
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

// Returns true (and fills hg) when the descriptor is a dense 3x3 / stride-1 conv whose 256-pixel tiles
// are whole image rows or whole images, and the halo tile fits in LDS next to the weight ring.
static bool halo_geom_for(const nbdt_conv_desc* d, int tile, int nwv, HaloGeom* hg) {
  const int gw = d->gw, gh = d->gh;
  if (gw > tile || tile % gw != 0) return false;
  int ib, rb;
  if (gw * gh >= tile) {
    rb = tile / gw;
    if (gh % rb != 0) return false;
    ib = 1;
  } else {
    if (tile % (gw * gh) != 0) return false;
    ib = tile / (gw * gh);
    rb = gh;
  }
  hg->ib = ib; hg->rb = rb;
  hg->hw2 = gw + 2;
  hg->himg = (rb + 2) * (gw + 2);
  hg->hp = ib * hg->himg;
  const int instr = (hg->hp * 4 + 63) / 64;
  if ((instr + nwv - 1) / nwv > (nwv == 4 ? 10 : 7) || instr < nwv) return false;
  hg->a_instr = instr;
  hg->a_bytes = instr * 1024;
  hg->blocks_per_img = ib == 1 ? gh / rb : 1;
  hg->nwv = nwv;
  const int nt32 = d->cout / 32;
  const int nt = nt32 % 5 == 0 ? 5 : (nt32 % 4 == 0 ? 4 : (nt32 % 2 == 0 ? 2 : 1));
  const int lds = 2 * hg->a_bytes + nw_slots(nwv) * nt * 32 * BK * 2;
  // 4 waves: 2 blocks per CU (80 KiB each); 8 waves: 1 block per CU
  return lds <= (nwv == 4 ? 80 : 156) * 1024;
}

// Returns true (and fills hg) when the descriptor is a dense 3x3 / stride-1 conv whose pixel tiles are
// whole image rows or whole images, and the halo tile fits in LDS next to the weight ring.  Prefers the
// 256-pixel / 4-wave form: alone it is 0.8 % slower per training step than the 512-pixel / 8-wave one (which
// halves the weight-tile traffic per MFMA), but two of its blocks -- or one of them and a weight-gradient
// block running on the engine's second stream -- share a CU, which is worth 2 % more.  The caller's
// `wide_tile` hint (forward launches) selects the wide tile; NBDT_HALO_W8=1 / NBDT_HALO_W4=1 force one.
bool conv_halo_applicable(const nbdt_conv_desc* d, int M, HaloGeom* hg) {
  if (d->ntaps != 9 || d->in_base != 0 || d->accumulate) return false;
  if (d->in_ws != d->cin || d->in_hs != (d->gw + 2) * d->cin || d->in_bs != (d->gh + 2) * d->in_hs) return false;
  for (int t = 0; t < 9; ++t)
    if (d->tap_off[t] != (t / 3) * d->in_hs + (t % 3) * d->in_ws) return false;
  static const bool force8 = getenv("NBDT_HALO_W8") != nullptr, force4 = getenv("NBDT_HALO_W4") != nullptr;
  const bool w4 = force4 || !(force8 || d->wide_tile);
  // the wide tile needs enough tiles to fill the chip (1 block per CU)
  const long long tiles512 = ((long long)M + 511) / 512 * (d->cout / (32 * (d->cout / 32 % 5 == 0 ? 5 : (d->cout / 32 % 4 == 0 ? 4 : (d->cout / 32 % 2 == 0 ? 2 : 1)))));
  if (!w4 && tiles512 >= 256 && halo_geom_for(d, 512, 8, hg)) return true;
  return halo_geom_for(d, 256, 4, hg);
}

int conv3x3_halo(const nbdt_conv_desc* d, const HaloGeom& hg, const void* in, const void* w, void* out,
                 const void* res, float* stats, const BnBwdArgs* bn, int M, hipStream_t st) {
  ConvDmaParams p;
  p.d = *d;
  p.in = (const bf16_t*)in;
  p.w = (const bf16_t*)w;
  p.w_tiled = nullptr;
  static const bool no_tiled = getenv("NBDT_NO_WTILED") != nullptr;   // A/B switch
  if (d->w_tiled != 0 && !no_tiled) {   // only with the identity tap map (forward weights / already tap-reversed dgrad copy)
    bool ident = d->w_ntaps == 9;
    for (int t = 0; t < 9; ++t) ident = ident && d->w_tap[t] == t;
    if (ident) p.w_tiled = (const bf16_t*)(uintptr_t)d->w_tiled;
  }
  p.out = (bf16_t*)out;
  p.res = (const bf16_t*)res;
  p.stats = stats;
  p.bn_x = bn ? (const bf16_t*)bn->x : nullptr;
  p.bn_mean = bn ? bn->mean : nullptr; p.bn_rstd = bn ? bn->rstd : nullptr;
  p.bn_gamma = bn ? bn->gamma : nullptr; p.bn_beta = bn ? bn->beta : nullptr;
  p.aff_scale = bn ? bn->aff_scale : nullptr; p.aff_shift = bn ? bn->aff_shift : nullptr;
  p.aff_act = bn ? bn->aff_act : 0;
  p.M = M;
  p.debug = 0;
  const int nt32 = d->cout / 32;
  if (hg.nwv == 8) {
    if (nt32 % 5 == 0) return launch_halo<5, 8>(p, hg, st);
    if (nt32 % 4 == 0) return launch_halo<4, 8>(p, hg, st);
    if (nt32 % 2 == 0) return launch_halo<2, 8>(p, hg, st);
    return launch_halo<1, 8>(p, hg, st);
  }
  if (nt32 % 5 == 0) return launch_halo<5, 4>(p, hg, st);
  if (nt32 % 4 == 0) return launch_halo<4, 4>(p, hg, st);
  if (nt32 % 2 == 0) return launch_halo<2, 4>(p, hg, st);
  return launch_halo<1, 4>(p, hg, st);
}

}  // namespace nbdt
