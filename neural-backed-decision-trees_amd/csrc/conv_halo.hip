// 3x3 / stride-1 / pad-1 implicit GEMM with an LDS-resident HALO tile (gfx950 only).
//
// Why a halo tile: ablation of conv_igemm_dma_kernel on MI355X (B=512, 32x32, 160->160) showed the kernel
// pulling 2.4 GB per launch through L2 because every one of the 9 taps re-loaded its (shifted) input tile.
// Here a block loads, per 32-channel slice, the input rows it needs INCLUDING the one-pixel halo ONCE and the
// 9 taps read shifted windows of that tile straight from LDS: tap (r,s) is a constant offset of r*(W+2)+s halo
// pixels for every lane.  The halo of neighbouring tiles / images is the zero border of the padded NHWC layout,
// so there is no predication.  In that layout a tile's halo -- (rows+2) x (W+2) pixels of one image, or whole
// padded images -- is ONE CONTIGUOUS run of padded pixels, so every DMA source address is base + piece * const.
//
// Two kernels share the tile geometry, LDS swizzle, MFMA mapping (conv_common.h) and epilogue:
//
//   conv3x3_pp_kernel   512 pixels x 32*NT couts, 8 waves, 1 block per CU -- the production kernel.
//       "Ping-pong": the two waves that share a SIMD (wave w and w+4) alternate roles every half step.  While
//       group A issues the 20 MFMAs of K step t (640 matrix-pipe cycles), group B -- one barrier behind --
//       reads the 14 operand fragments of its step from LDS, waits for last step's LDS-DMA, issues its share of
//       the next weight tile / halo slice and drains lgkmcnt; then they swap.  Round 1's kernel had every wave
//       do  wait -> barrier -> issue DMA -> reads+MFMAs  in lockstep: both waves of a SIMD stalled together on
//       DMA issue and LDS latency (ablation: MFMA-only 125 us, +LDS reads 165, +barriers 182, +DMA 263 us per
//       launch).  With the roles split a SIMD's matrix pipe always has one wave in a pure-MFMA segment.
//   conv3x3_halo_kernel 256 pixels x 32*NT couts, 4 waves, 2 blocks per CU -- round 1's structure, kept for
//       grids too small to give every CU a 512-pixel tile (small batches, tests).
//
// K steps t = (kc, tap), tap fastest; per step a block consumes one 32-channel weight tile W(t) (ring of 3
// slots) and a window of halo slice A(kc) (2 buffers).
#include "conv_common.h"
#include <cmath>
#include <map>
#include <mutex>

#include <algorithm>

// counted wait: the `n` most recent DMA instructions of this wave may stay in flight (immediate operand)
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
#define NBDT_CASE(K) case K: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K) : "memory"); break;
    NBDT_CASE(1) NBDT_CASE(2) NBDT_CASE(3) NBDT_CASE(4) NBDT_CASE(5) NBDT_CASE(6) NBDT_CASE(7) NBDT_CASE(8)
    NBDT_CASE(9) NBDT_CASE(10) NBDT_CASE(11) NBDT_CASE(12) NBDT_CASE(13) NBDT_CASE(14) NBDT_CASE(15)
    NBDT_CASE(16) NBDT_CASE(17) NBDT_CASE(18)
#undef NBDT_CASE
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
  }
}

constexpr int min_w_dma_h(int w_instr, int nwv) {
  int best = 1 << 30;
  for (int w = 0; w < nwv; ++w) {
    int n = 0;
    for (int id = w; id < w_instr; id += nwv) ++n;
    best = n < best ? n : best;
  }
  return best;
}

// What a block needs to know about its tile, in SGPRs: the padded-pixel index of the first halo pixel and the
// clamp for reads past the end of the tensor (tail lanes of the last halo piece, images past the batch).
struct TileOrigin {
  int base_pix;     // padded pixel index (over [B][H+2][W+2]) of halo pixel 0
  int last_pix;     // B*(H+2)*(W+2) - 1
};
__device__ __forceinline__ TileOrigin tile_origin(const nbdt_conv_desc& d, const nbdt::HaloGeom& hg, int m_blk) {
  const int img_pp = (d.gh + 2) * (d.gw + 2);
  TileOrigin o;
  if (hg.ib == 1) {
    const int b0 = m_blk / hg.blocks_per_img;
    o.base_pix = b0 * img_pp + (m_blk - b0 * hg.blocks_per_img) * hg.rb * (d.gw + 2);
  } else {
    o.base_pix = m_blk * hg.ib * img_pp;
  }
  o.last_pix = d.B * img_pp - 1;
  return o;
}

#ifndef NBDT_PP_SCHED
#define NBDT_PP_SCHED 0    // schedule experiments: 1 no s_setprio around the MFMA segment, 2 no MFMA/VALU interleave
                           // directives, 4 per-block rotation of the weight pieces
#endif
#ifndef NBDT_PP_TIMING
#define NBDT_PP_TIMING 0   // 1: s_memtime stamps around the segments of every step, per-wave sums in g_pp_timing
                           // 2: four stamps per block + HW_ID / XCC_ID (scratch/pp_trace.py: a CU's timeline)
#endif
#if NBDT_PP_TIMING
__device__ unsigned g_pp_epi[8192 * 8];        // [block*8 + wave][8]: epilogue phases
extern "C" int nbdt_debug_pp_epi(unsigned* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pp_epi), sizeof(unsigned) * 8192 * 8);
}
__device__ unsigned g_pp_timing[8192 * 8];     // [block*8 + wave][8]: load segment, barrier 1, MFMA segment, barrier 2, total, steps
extern "C" int nbdt_debug_pp_timing(unsigned* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pp_timing), sizeof(unsigned) * 8192 * 8);
}
__device__ __forceinline__ unsigned stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return (unsigned)t;
}
#endif
#ifndef NBDT_PP_ABLATE
#define NBDT_PP_ABLATE 0   // compile-time timing experiments (scratch/ablate_pp.sh): 1 no DMA, 2 no halo DMA after the
                           // prologue, 4 no MFMA, 8 no LDS fragment reads, 32 no epilogue stores
#endif

// ------------------------------------------------------------------------------------------------------------
// 8-wave ping-pong kernel.  Barrier timeline (b = s_barrier, L/M = load / MFMA segment of a K step):
//   group 0 (waves 0-3):  bP  L0 b  M0 b  L1 b  M1 b ...  L(n-1) b  M(n-1) b  b
//   group 1 (waves 4-7):  bP  b  L0 b  M0 b  L1 b  M1 ...           L(n-1) b  M(n-1) b
// L(t): 4+2*NT ds_read_b128 of step t | vmcnt: this wave's W pieces of L(t-1) have landed | issue this wave's
//       pieces of W(t+2), then one piece of A(kc+1) | lgkmcnt(0).
// Hazards: W(t+2) reuses the slot of W(t-1), whose last reader (group 1, L(t-1)) drained lgkmcnt before the
// barrier that precedes group 0's L(t).  W(t+2) is waited for by every issuing wave in its L(t+1), i.e. before
// the barrier that precedes the first reader (group 0, L(t+2)).  A(kc+1) overwrites the buffer of A(kc-1),
// dead since step 9kc-1; its pieces are issued at taps 0..6 of slice kc and retired by the wait two steps later
// (tap 6 -> the vmcnt(0) of tap 8), a barrier before step 9kc+9 reads them.  Nothing but vmcnt + a barrier orders LDS-DMA against ds_read.
//
// LDS pitch (round 5, PAD = true; images narrower than 32 pixels).  A ds_read_b128 is served in passes of 16 lanes --
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same in the upper half -- and the 16-byte chunk swizzle
// c ^ ((p >> 2) & 3) makes a pass conflict-free when its 16 pixels are DISTINCT MOD 16, which consecutive pixels of the
// passes' lane sets are.  A 32-pixel fragment is one row of a 32-wide image but two rows of a 16-wide and four of an
// 8-wide one, and in a halo copied as the contiguous run it is in memory the row pitch is W + 2: lanes 20-27 sit at
// p + 22 .. p + 29, which collides with lanes 12-15 -- SQ_LDS_BANK_CONFLICT 24 % (16x16) and 37 % (8x8) of the LDS-active
// cycles (profiles/r04_lds_conflicts_by_stage.txt).  With W + 2 = 2 (mod 4) no swizzle fixes that (the physical quarter
// p & 3 flips between rows: DESIGN 5.3).  So the LDS image gets rows of W + 4 slots -- the DMA skips two slots per row;
// what lands in them is never read -- and the swizzle is taken on the DE-PITCHED coordinate v = column + W * row: the
// physical quarter is column & 3 in every row, a fragment's pixels are consecutive in v whatever the tap, and the
// original argument holds again: zero conflicts at every width.  Costs: 11 % (16x16) / 20 % (8x8) more LDS for the two
// halo buffers -- 8x8x8-image tiles need 60 pieces per slice where taps 0..6 give 56 issue slots: waves 0-3 issue a second
// piece at tap 0 -- and ~8 VALU per K step in the MFMA segment's shadow (a slot's row by multiply-shift, packed (slot, v)
// fragment coordinates).  PAD = false is the kernel of rounds 2-4, instruction for instruction (32-wide images).
// MW = 32-pixel fragments per wave: 2 (64 pixels, tiles of 64 * NWV) or, 8 waves only, 1 -- the HALF tile of 256 pixels
// for grids that give 512-pixel tiles to fewer than 3/4 of the CUs: the same two wave groups and segments, a wave's MFMA
// segment is 2 * NT instead of 4 * NT instructions, every CU gets a tile (round 5).
template <int NT, bool HAS_RES, int STATS, int NWV, bool PAD, int MW = 2>
__global__ __launch_bounds__(64 * NWV, 2) void conv3x3_pp_kernel(nbdt::ConvDmaParams p, nbdt::HaloGeom hg) {
  static_assert(MW == 2 || (MW == 1 && NWV == 8 && !PAD), "half tiles are the 8-wave kernel's");
  constexpr bool PP = NWV == 8;          // two wave groups alternating roles; NWV == 4: one group, two blocks per CU
  constexpr int APW = PP ? 1 : 2;        // halo pieces a wave may issue per step (taps 0..6: 7 * NWV * APW slots)
  constexpr int BN = 32 * NT;
  constexpr int BMH = 32 * MW * NWV;
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int W_INSTR = W_BYTES / 1024;
  constexpr int abl = NBDT_PP_ABLATE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [A buf 0][A buf 1][W ring x3]
#if NBDT_PP_TIMING == 2
  unsigned tr_entry = stamp();   // entry of the block, then the end of the previous tile
#endif

  // Persistent blocks: block b sits on XCD b & 7 (round-robin dispatch) and walks that XCD's contiguous item
  // range with the XCD's other blocks, items  xcd * per_xcd + (b >> 3) + j * (gridDim / 8).
  const int bid = blockIdx.x;
  const int item_step = gridDim.x >> 3;
  constexpr bool KSPLIT = MW == 1;       // half tiles only: several blocks per output tile, each a range of K slices
  const int ksplit = KSPLIT ? __builtin_amdgcn_readfirstlane(p.ksplit) : 1;
  const int item_end = min(((bid & 7) + 1) * p.per_xcd, p.m_blocks * p.n_blocks * ksplit);
  int item = (bid & 7) * p.per_xcd + (bid >> 3);
  if (item >= item_end) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = PP ? wave >> 2 : 0;
  const nbdt_conv_desc& d = p.d;

#define NBDT_PIN(x) __builtin_amdgcn_readfirstlane(x)
  const int cin = NBDT_PIN(d.cin);
  const int kchunks_w = cin >> 5;            // K chunks the weight tiles were laid out for
#ifdef NBDT_PP_KFRAC5       // timing experiment (scratch/variants): only KFRAC5/5 of the K loop -- a main loop that much faster
  const int kchunks_all = kchunks_w * NBDT_PP_KFRAC5 / 5;
#else
  const int kchunks_all = kchunks_w;
#endif
  int kchunks = kchunks_all;                 // K slices of the CURRENT tile (split K: this block's range of them)
  int nk = 9 * kchunks;
  const int a_bytes = NBDT_PIN(hg.a_bytes);
  const int a_instr = NBDT_PIN(hg.a_instr);
  const int hw2 = NBDT_PIN(hg.hw2), himg = NBDT_PIN(hg.himg);
  const int lpitch = NBDT_PIN(hg.lpitch), limg = NBDT_PIN(hg.limg), row_magic = NBDT_PIN(hg.row_magic);   // (PAD only)
  const unsigned long long in_u = (unsigned long long)p.in, w_u = (unsigned long long)p.w_tiled;
  const bf16_t* in_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(in_u >> 32)) << 32) |
                                          (unsigned)NBDT_PIN((unsigned)in_u));
  const bf16_t* w_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(w_u >> 32)) << 32) |
                                         (unsigned)NBDT_PIN((unsigned)w_u));
  struct Tile {
    int item, m_blk, n_blk, m0, n0, base_pix;
    int tile, ks, nkc;         // output tile, this block's K split and its number of slices
    const bf16_t* w_tiles;     // this cout tile's DMA-ordered weight tiles, this block's first step
    const bf16_t* in_k;        // input tensor at this block's first slice
  };
  auto tile_of = [&](int it) {
    Tile t;
    t.item = it;
    t.tile = it; t.ks = 0;
    if (KSPLIT && ksplit > 1) { t.tile = it / ksplit; t.ks = it - t.tile * ksplit; }
    t.m_blk = t.tile / p.n_blocks;
    t.n_blk = t.tile - t.m_blk * p.n_blocks;
    t.m0 = t.m_blk * BMH;
    t.n0 = t.n_blk * BN;
    t.base_pix = NBDT_PIN(tile_origin(d, hg, t.m_blk).base_pix);
    const int kc0 = NBDT_PIN((t.ks * kchunks_all) / ksplit);
    t.nkc = NBDT_PIN(((t.ks + 1) * kchunks_all) / ksplit) - kc0;
    t.w_tiles = w_base + ((size_t)t.n_blk * kchunks_w + kc0) * 9 * (BN * 32);
    t.in_k = in_base + kc0 * BK;
    return t;
  };
  Tile cur = tile_of(item);
  const int last_pix = NBDT_PIN(d.B * (d.gh + 2) * (d.gw + 2) - 1);
#undef NBDT_PIN

  // ---- DMA source addressing: every LDS-DMA is  global_load_lds  <32-bit lane offset>, <SGPR base>  -- the
  // wave-uniform part of an address (slice, tap, piece, tile) is scalar arithmetic, the lane part one VGPR.
  // A piece `id` = halo pixels [16 id, 16 id + 16) x 64 B; lane -> pixel lane>>2, LDS chunk position lane&3,
  // which holds source chunk (lane&3) ^ swizzle(pixel); 16 | 16*id so the swizzle (pixel>>2)&3 is per lane.
  // (the per-lane constants below are recomputed at the top of every tile from an opaque copy of `lane`: kept
  // live across the epilogue they cost the accumulator pass registers it does not have)
  int a_lane_pix, a_lane_el;                          // (PAD: a_lane_el holds the chunk POSITION lane & 3)
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // PAD: byte offset of what LDS slot x = 16 id + (lane >> 2), chunk position lane & 3, must hold.  Slot x is column
  // x - R * lpitch of halo row R = x / lpitch (rows of all the tile's images, stacked); the halo in memory has rows of
  // lpitch - 2 pixels, so its pixel is base + x - 2 R, and the chunk is the position XOR the swizzle of v = x - 4 R.
  auto pad_voff = [&](int x, int lc, int pix_base) {
    const int R = (x * row_magic) >> 16;
    int px = x - 2 * R + pix_base;
    px = px < last_pix ? px : last_pix;                // tail lanes / images past the batch re-read the last pixel
    const int v = x - 4 * R;
    return (unsigned)(px * cin + ((lc ^ ((v >> 2) & 3)) << 3)) * 2u;
  };
  auto issue_a_piece = [&](const bf16_t* src, int base_pix, int buf, int id) {      // src: the tensor at the slice to fetch
    int lp = a_lane_pix, le = a_lane_el;
    asm volatile("" : "+v"(lp), "+v"(le));            // keep this address math inside the step (registers)
    if (PAD) {
      glds16_sf(src, pad_voff(lp + id * 16, le, base_pix), lds_base + buf * a_bytes + id * 1024);
      return;
    }
    int px = lp + (base_pix + id * 16);
    px = px < last_pix ? px : last_pix;                // tail lanes / images past the batch re-read the last pixel
    glds16_s(src, (unsigned)(px * cin + le) * 2u, lds_base + buf * a_bytes + id * 1024);
  };
  // W piece `id` = rows [16 id, 16 id + 16) of a weight tile.  The weights are the DMA-ordered tiles of
  // nbdt_weight_tile_batched: tile (n_blk, kc, tap) IS the swizzled LDS image, stored contiguously in step order,
  // so piece id of step t is  w_tiles + (t * W_INSTR + id) KiB  and every lane reads 16 B at 16 * lane.
  unsigned w_voff;
  const unsigned w_ring = lds_base + 2 * a_bytes;
  // Blocks run in lockstep (same start, same step time): without the rotation every CU of an XCD would ask its L2
  // for the SAME KiB at the same moment.  Block b starts W_ROT pieces further into the tile.
  const int w_rot = (NBDT_PP_SCHED & 4) ? item % W_INSTR : 0;
  constexpr int IPW = (W_INSTR + NWV - 1) / NWV;
  auto issue_w = [&](const bf16_t* w_tiles, int slot, int t) {
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
      const int slot_id = wave + NWV * k;
      if (slot_id < W_INSTR) {   // wave-uniform
        int id = slot_id + w_rot;
        id = id >= W_INSTR ? id - W_INSTR : id;
        glds16_s(w_tiles + (t * W_INSTR + id) * 512, w_voff, w_ring + slot * W_BYTES + id * 1024);
      }
    }
  };

  f32x16 acc[NT][MW];
  int frag_half;
  int hp0[MW];                     // halo index of this lane's two output pixels at tap (0,0)
  unsigned w_rd0, w_rd1;           // LDS offsets of this lane's weight fragment rows (ks = 0, 1) in ring slot 0
  auto lane_constants = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    a_lane_pix = ln >> 2;
    a_lane_el = PAD ? (ln & 3) : (((ln & 3) ^ ((ln >> 4) & 3)) << 3);
    w_voff = ln * 16u;
    const int frag_row = ln & 31;
    frag_half = ln >> 5;
#pragma unroll
    for (int tm = 0; tm < MW; ++tm) {
      const int pl = wave * (32 * MW) + tm * 32 + frag_row;
      const int per_img = hg.rb * d.gw;
      const int img = pl / per_img;
      const int rem = pl - img * per_img;
      const int r = rem / d.gw, c = rem - r * d.gw;
      // PAD: LDS slot | swizzle coordinate << 16 (both < 2048: 8 x 10 x 12 = 960 slots at most)
      hp0[tm] = PAD ? ((img * limg + r * lpitch + c) | ((img * limg + r * lpitch + c - 4 * (img * (hg.rb + 2) + r)) << 16))
                    : img * himg + r * hw2 + c;
    }
    const int w_frag_off = frag_row * 64 + ((frag_half ^ ((frag_row >> 2) & 3)) << 4);   // ks = 0; ks = 1 is ^ 32
    w_rd0 = 2 * a_bytes + w_frag_off;
    w_rd1 = 2 * a_bytes + (w_frag_off ^ 32);
  };
  lane_constants();

  // ---- everything a load segment needs is prepared one segment EARLIER, inside the previous MFMA segment: a wave
  // issues one instruction per ~4 cycles, a load segment that also computed its 4 swizzled LDS addresses and its
  // DMA operands was ~100 instructions = 840 cycles long (s_memtime stamps, profiles/r02_pp_segments.txt) and set
  // the pace, while the MFMA segment used 20 of its 160 issue slots.  Now L(t) = reads, wait, <= 3 DMA, wait.
  typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;   // 32-bit: reads stay ds_read_b128
  const lds_cptr smem3 = (lds_cptr)smem;
  struct Plan {
    lds_cptr ra[2][MW];       // LDS addresses of the pixel fragments [ks][tm]
    unsigned a_voff[APW];     // halo pieces: per-lane byte offsets
    int a_n;                  // ... and how many of them this wave issues in L(t)
    unsigned a_xoff;          // PAD: piece 56 + wave of the next slice, issued at tap 0 when the slice has more than
    int a_x;                  //      the 56 pieces taps 0..6 have slots for (8x8 images: 60)
  };
  int a_pix0 = cur.base_pix + wave * 16;       // halo pixel of lane 0 of this wave's piece at tap 0 (per tile)
  auto prepare = [&](int tapn, int kcn) {     // plan of L(t) for t = (kcn, tapn); tapn is a literal after unrolling
    Plan q;
    int h0 = hp0[0], h1 = hp0[MW - 1], lp = a_lane_pix, le = a_lane_el;
    asm volatile("" : "+v"(h0), "+v"(h1), "+v"(lp), "+v"(le));    // the address math stays in the segment that calls prepare()
    // tap offset: halo pixels; PAD: LDS slots | de-pitched pixels << 16 (rows of lpitch slots / lpitch - 4 pixels)
    const int toff = PAD ? (((tapn / 3) * lpitch + (tapn % 3)) | (((tapn / 3) * (lpitch - 4) + (tapn % 3)) << 16))
                         : (tapn / 3) * hw2 + (tapn % 3);
    const unsigned abuf = (kcn & 1) * a_bytes;
    auto frag_off = [&](int h) {
      const int hp = h + toff;
      if (PAD) return (unsigned)(((hp << 6) & 0x3fffc0) + ((frag_half ^ ((hp >> 18) & 3)) << 4));
      return (unsigned)(hp * 64 + ((frag_half ^ ((hp >> 2) & 3)) << 4));
    };
    {
      const unsigned o = frag_off(h0);
      q.ra[0][0] = smem3 + (abuf + o); q.ra[1][0] = smem3 + (abuf + (o ^ 32));
    }
    if constexpr (MW == 2) {
      const unsigned o = frag_off(h1);
      q.ra[0][MW - 1] = smem3 + (abuf + o); q.ra[1][MW - 1] = smem3 + (abuf + (o ^ 32));
    }
    // up to APW pieces of the next halo slice per wave at taps 0..6: piece id = (tapn * APW + j) * NWV + wave
    q.a_n = 0;
#pragma unroll
    for (int j = 0; j < APW; ++j) {
      const int id0 = (tapn * APW + j) * NWV;            // literal
      if (tapn < 7 && kcn + 1 < kchunks && id0 + wave < a_instr) q.a_n = j + 1;
      if (PAD) {
        q.a_voff[j] = pad_voff(lp + (id0 + wave) * 16, le, a_pix0 - wave * 16);
        continue;
      }
      int px = lp + (a_pix0 + id0 * 16);
      px = px < last_pix ? px : last_pix;     // tail lanes / images past the batch re-read the last pixel
      q.a_voff[j] = (unsigned)(px * cin + le) * 2u;
    }
    q.a_x = 0;
    q.a_xoff = 0;
    if (PAD && tapn == 0) {
      q.a_x = (kcn + 1 < kchunks && 7 * APW * NWV + wave < a_instr) ? 1 : 0;
      q.a_xoff = pad_voff(lp + (7 * APW * NWV + wave) * 16, le, a_pix0 - wave * 16);
    }
    return q;
  };

  // ---- a tile's first LDS-DMA: A(0) (every piece) into halo buffer 0, W(0), W(1) into ring slots 0, 1
  auto issue_first = [&](const Tile& t) {
    if (abl & 1) return;
#pragma unroll
    for (int k = 0; k < (PAD ? 8 : 7) * APW; ++k)
      if (wave + NWV * k < a_instr) issue_a_piece(t.in_k, t.base_pix, 0, wave + NWV * k);
    issue_w(t.w_tiles, 0, 0);
    issue_w(t.w_tiles, 1, 1);
  };
  // ---- LDS of the epilogue.  Packed from smem + 0 it overlaps halo buffer 0 and the ring, so the next tile
  // cannot be started before it is done.  With p.overlap (the launch checked that it fits) the per-wave regions
  // go into halo buffer 1 and behind ring slot 1 instead, and the next tile's first DMA is issued at the top of
  // the epilogue: its HBM round trip (10 k of 81 k cycles per tile at 32x32x160, profiles/r02_pp_trace.txt) and
  // the dispatch gap between two blocks (2.4 k) disappear behind the 8.8 k-cycle epilogue.
  constexpr int EPI_REGION = 32 * (2 * BN + 16);
  EpiLds epi_lds = epi_lds_packed<NT, NWV>(smem, wave);
  if (p.overlap) {
    const int in_a1 = a_bytes / EPI_REGION;                 // regions that fit into halo buffer 1
    unsigned char* tail = smem + 2 * a_bytes + 2 * W_BYTES;  // ring slot 2 and everything behind it
    const int n_tail = NWV - (in_a1 < NWV ? in_a1 : NWV);
    epi_lds.region = wave < in_a1 ? smem + a_bytes + wave * EPI_REGION : tail + (wave - in_a1) * EPI_REGION;
    epi_lds.row_off = (int*)(tail + n_tail * EPI_REGION) + wave * 64;
    epi_lds.blk_stats = (float*)(tail + n_tail * EPI_REGION + NWV * 64 * 4);
    epi_lds.base_a = smem + a_bytes; epi_lds.base_b = tail; epi_lds.n_a = in_a1 < NWV ? in_a1 : NWV;
  }

  issue_first(cur);
  for (;;) {   // ======================================= one output tile =======================================
  const int m_blk = cur.m_blk, m0 = cur.m0, n0 = cur.n0;
  const bf16_t* w_tiles = cur.w_tiles;
  const bf16_t* in_k = cur.in_k;
  item = cur.item;
  if (KSPLIT) { kchunks = cur.nkc; nk = 9 * kchunks; }
  lane_constants();
  a_pix0 = cur.base_pix + wave * 16;
#pragma unroll
  for (int tn = 0; tn < NT; ++tn)
#pragma unroll
    for (int tm = 0; tm < MW; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;
  Plan plan = prepare(0, 0);
  int prev_a = 0;                          // halo pieces this wave issued in the previous load segment
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's first DMA (and the last tile's stores)
  __builtin_amdgcn_s_barrier();            // bP: every wave's pieces have landed; the last epilogue's LDS is free
  if (PP && grp == 1) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

#if NBDT_PP_TIMING == 2
  const unsigned tr_loop = stamp();
#define NBDT_STAMP(acc_)
#elif NBDT_PP_TIMING
  unsigned tm_l = 0, tm_b1 = 0, tm_m = 0, tm_b2 = 0;
  const unsigned tm_begin = stamp();
  unsigned tm_prev = tm_begin;
#define NBDT_STAMP(acc_) { const unsigned t_ = stamp(); acc_ += t_ - tm_prev; tm_prev = t_; }
#else
#define NBDT_STAMP(acc_)
#endif
  for (int kc = 0; kc < kchunks; ++kc) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // ================= L(t): fragments -> registers, this wave's DMA pieces =================
      bf16x8 pf[2][MW], wf[2][NT];
      if (!(abl & 8)) {
#pragma unroll
        for (int tm = 0; tm < MW; ++tm) {
          pf[0][tm] = *(const __attribute__((address_space(3))) bf16x8*)plan.ra[0][tm];
          pf[1][tm] = *(const __attribute__((address_space(3))) bf16x8*)plan.ra[1][tm];
        }
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
          wf[0][tn] = *(const bf16x8*)(smem + w_rd0 + ((tap % 3) * W_BYTES + tn * 2048));
          wf[1][tn] = *(const bf16x8*)(smem + w_rd1 + ((tap % 3) * W_BYTES + tn * 2048));
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int tm = 0; tm < MW; ++tm) { pf[ks][tm] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; asm volatile("" : "+v"(pf[ks][tm])); }
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) { wf[ks][tn] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; asm volatile("" : "+v"(wf[ks][tn])); }
        }
      }
      // Wait for the weight pieces this wave issued one step ago.  A halo piece issued in that step came LAST in
      // issue order and is not needed before the next slice: it may stay in flight (vmcnt counts in order), so
      // its HBM latency is never exposed; the next step's wait retires it.
      if (prev_a == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (prev_a == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      if (!(abl & 1)) {
        const int t2 = kc * 9 + tap + 2;                    // W(t+2) -> ring slot (t+2) % 3 = (tap+2) % 3
        if (t2 < nk) issue_w(w_tiles, (tap + 2) % 3, t2);
        if (!(abl & 2)) {                                   // pieces (tap*APW + j)*NWV + wave of slice kc+1
#pragma unroll
          for (int j = 0; j < APW; ++j)
            if (j < plan.a_n)
              glds16_s(in_k + (kc + 1) * BK, plan.a_voff[j],
                       lds_base + ((kc + 1) & 1) * a_bytes + ((tap * APW + j) * NWV + wave) * 1024);
          if (PAD && tap == 0 && plan.a_x)
            glds16_sf(in_k + (kc + 1) * BK, plan.a_xoff,
                      lds_base + ((kc + 1) & 1) * a_bytes + (7 * APW * NWV + wave) * 1024);
        }
      }
      prev_a = (abl & 3) ? 0 : plan.a_n + ((PAD && tap == 0) ? plan.a_x : 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      NBDT_STAMP(tm_l)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      NBDT_STAMP(tm_b1)
      // ================= M(t): 4*NT MFMAs; the idle issue slots between them prepare L(t+1) =================
      if (!(NBDT_PP_SCHED & 1)) __builtin_amdgcn_s_setprio(1);
      plan = tap < 8 ? prepare(tap + 1, kc) : prepare(0, kc + 1);     // (after the last step: computed, never used)
      if (!(abl & 4)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int tm = 0; tm < MW; ++tm)
              acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][tn], pf[ks][tm], acc[tn][tm], 0, 0, 0);
      }
#ifdef NBDT_PP_DUMMY_VALU   // experiment: how much does extra VALU work in the MFMA shadow cost (an in-LDS BatchNorm pass)?
      {
        float d0 = __int_as_float(lane), d1 = d0 + 1.f, d2 = d0 + 2.f, d3 = d0 + 3.f;
#pragma unroll
        for (int i = 0; i < NBDT_PP_DUMMY_VALU / 4; ++i) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d0) : "v"(d1), "v"(d2));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d1) : "v"(d2), "v"(d3));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d2) : "v"(d3), "v"(d0));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d3) : "v"(d0), "v"(d1));
        }
        asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3));
      }
#endif
      // one MFMA, then at most two of the preparation's VALU / SALU instructions in its shadow, 20 times
      if (!(NBDT_PP_SCHED & 2)) {
#pragma unroll
        for (int i = 0; i < 2 * MW * NT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
#ifdef NBDT_PP_DUMMY_VALU
          __builtin_amdgcn_sched_group_barrier(0x002, 2 + NBDT_PP_DUMMY_VALU / 20 + 1, 0);   // VALU
#else
          __builtin_amdgcn_sched_group_barrier(0x002, MW == 2 ? 2 : 3, 0);   // VALU (half tile: the same preparation, half the MFMAs)
#endif
          __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);   // SALU
        }
      }
      asm volatile("" : "+v"(plan.ra[0][0]), "+v"(plan.ra[1][0]), "+v"(plan.a_voff[0]));
      if (MW == 2) asm volatile("" : "+v"(plan.ra[0][MW - 1]), "+v"(plan.ra[1][MW - 1]));
      if (APW == 2) asm volatile("" : "+v"(plan.a_voff[APW - 1]));
      if (PAD && tap == 8) asm volatile("" : "+v"(plan.a_xoff));
      if (!(NBDT_PP_SCHED & 1)) __builtin_amdgcn_s_setprio(0);
      NBDT_STAMP(tm_m)
      __builtin_amdgcn_sched_barrier(0);
      if (PP) __builtin_amdgcn_s_barrier();      // (one group: L(t+1) only needs what the barrier after L(t) ordered)
      __builtin_amdgcn_sched_barrier(0);
      NBDT_STAMP(tm_b2)
    }
  }
#if NBDT_PP_TIMING == 2
  const unsigned tr_epi = stamp();
#elif NBDT_PP_TIMING
  if (lane == 0 && item < 1024) {
    unsigned* o = g_pp_timing + (item * 8 + wave) * 8;
    o[0] = tm_l; o[1] = tm_b1; o[2] = tm_m; o[3] = tm_b2; o[4] = tm_prev - tm_begin; o[5] = nk;
  }
#endif
#undef NBDT_STAMP
  if (PP && grp == 0) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---- the next tile of this block, and when its first DMA can go out
  const int next_item = item + item_step;
  const bool more = next_item < item_end;
  const Tile nxt = more ? tile_of(next_item) : cur;
  const bool early = more && p.overlap && !(abl & 32);
  auto hook = [&]() {
    if (early) issue_first(nxt);
  };
  bool run_epilogue = true;
  if (KSPLIT && ksplit > 1) {
    // ---- split K: park this block's accumulators, take a ticket; the last block of the tile sums all ksplit partials in
    // split order (its own included: the result does not depend on who came last) and runs the epilogue.  Partials cross
    // XCDs, whose L2s are not coherent with each other: they are written and read with agent-scope (sc1) accesses, which
    // go through the L2 to memory.  (Not fences: __threadfence() is a write-back + invalidate of the WHOLE L2 per wave --
    // 256 blocks of them made a 39 us launch take 226 us.)
    constexpr int NQ = NT * MW * 4;                     // float4s per thread
    // (one running pointer, opaque to the optimiser: NQ hoisted 64-bit addresses would cost the K loop its registers)
    f32x4* mine = (f32x4*)p.k_ws + ((size_t)cur.tile * ksplit + cur.ks) * NQ * (64 * NWV) + tid;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int tm = 0; tm < MW; ++tm)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = f32x4{acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]};
          // (s_nop: wait states between a > 8-byte store issued from an asm string and a VALU write of its data registers --
          //  the next iteration's copy of the accumulators -- which hipcc's hazard recogniser cannot see; csrc/common.h st16)
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(mine), "v"(v) : "memory");
          mine += 64 * NWV;
          asm volatile("" : "+v"(mine));
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's partials have reached memory
    __builtin_amdgcn_s_barrier();                       // ... every wave's have; every wave is done with the K ring
    unsigned ticket = 0;
    if (tid == 0) ticket = atomicAdd(p.k_tickets + cur.tile, 1u);
    // (thread 0 sits in wave 0: the other waves get the ticket through LDS -- the K ring is free after the barrier above)
    volatile unsigned* tk = (volatile unsigned*)(smem + 2 * a_bytes);
    if (tid == 0) *tk = ticket;
    __syncthreads();
    ticket = *tk;
    __syncthreads();                                    // read before the next tile's first DMA may land there
    run_epilogue = ticket == (unsigned)(ksplit - 1);
    if (run_epilogue) {
      // ready for the next launch: an agent-scope atomic like the increments (a plain store beside sc1 atomics in the
      // same line relied on the end-of-kernel write-back for the next launch to see the zero -- ADVICE r5)
      if (tid == 0) __hip_atomic_store(p.k_tickets + cur.tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (agent-scope relaxed loads, 8 bytes each: the compiler keeps a whole split's worth in flight)
      const unsigned long long* all = (const unsigned long long*)((const f32x4*)p.k_ws + (size_t)cur.tile * ksplit * NQ * (64 * NWV) + tid);
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int tm = 0; tm < MW; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;
      for (int s2 = 0; s2 < ksplit; ++s2) {
        const unsigned long long* src = all + (size_t)s2 * NQ * (64 * NWV) * 2;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int tm = 0; tm < MW; ++tm)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              acc[tn][tm][4 * q] += __uint_as_float((unsigned)lo);
              acc[tn][tm][4 * q + 1] += __uint_as_float((unsigned)(lo >> 32));
              acc[tn][tm][4 * q + 2] += __uint_as_float((unsigned)hi);
              acc[tn][tm][4 * q + 3] += __uint_as_float((unsigned)(hi >> 32));
              src += 64 * NWV * 2;
            }
      }
    } else {
      hook();                                           // (what the epilogue would have done after its entry barrier)
    }
  }
#if NBDT_PP_TIMING == 2
  conv_epilogue<NT, HAS_RES, STATS, NWV, MW>(acc, p, epi_lds, m0, n0, m_blk, wave, lane, tid, nullptr, hook);
  if (tid == 0 && item < 8192) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned* o = g_pp_timing + item * 8;
    o[0] = hw; o[1] = xcc; o[2] = tr_entry; o[3] = tr_loop; o[4] = tr_epi; o[5] = stamp();
  }
  tr_entry = stamp();
#elif NBDT_PP_TIMING
  unsigned epi_t[8];
  conv_epilogue<NT, HAS_RES, STATS, NWV, MW>(acc, p, epi_lds, m0, n0, m_blk, wave, lane, tid, epi_t, hook);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stores acknowledged
  epi_t[6] = stamp();
  if (lane == 0 && item < 1024)
    for (int i = 0; i < 6; ++i) g_pp_epi[(item * 8 + wave) * 8 + i] = epi_t[i + 1] - epi_t[i];
#else
  if (!(abl & 32) && run_epilogue)
    conv_epilogue<NT, HAS_RES, STATS, NWV, MW>(acc, p, epi_lds, m0, n0, m_blk, wave, lane, tid, nullptr, hook);
#endif
  if (abl & 32) {   // timing experiment: no epilogue, but every accumulator stays live
    float sum = 0.f;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int tm = 0; tm < MW; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[tn][tm][r];
    if (sum == 12345.f) p.out[0] = 0;
  }
  if (!more) break;
  if (!early) {   // the packed epilogue LDS overlaps halo buffer 0 and the ring: every wave must be out of it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_first(nxt);
  }
  cur = nxt;
  }   // tile loop
}

// ------------------------------------------------------------------------------------------------------------
// 4-wave kernel (round 1 structure): per step  wait(vmcnt) -> s_barrier -> issue DMA -> reads + MFMAs.
//   A halo tile of slice kc+1 -> other A buffer, issued at (kc, tap 0) before W(t+2); W(t+2) -> ring slot (t+2)%3.
//   In-order completion of DMA loads makes "W(t) landed" imply "A(kc) landed" (issued >= 9 steps earlier).
template <int NT, bool HAS_RES, int STATS>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(nbdt::ConvDmaParams p, nbdt::HaloGeom hg) {
  constexpr int NWV = 4;
  constexpr int BN = 32 * NT;
  constexpr int BMH = 64 * NWV;
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int W_INSTR = W_BYTES / 1024;
  constexpr int IPW_W = (W_INSTR + NWV - 1) / NWV;
  constexpr int MINW = min_w_dma_h(W_INSTR, NWV);
  constexpr int MAX_A_SLOTS = 10;
  constexpr int NWS = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [A buf 0][A buf 1][W ring x3]

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
  const int hw2 = NBDT_PIN(hg.hw2), himg = NBDT_PIN(hg.himg);
  const bool tiled = NBDT_PIN(p.w_tiled != nullptr ? 1 : 0) != 0;
  const unsigned long long in_u = (unsigned long long)p.in, w_u = (unsigned long long)(tiled ? p.w_tiled : p.w);
  const bf16_t* in_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(in_u >> 32)) << 32) |
                                          (unsigned)NBDT_PIN((unsigned)in_u));
  const bf16_t* w_base = (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(w_u >> 32)) << 32) |
                                         (unsigned)NBDT_PIN((unsigned)w_u));
  const TileOrigin org = tile_origin(d, hg, m_blk);
  const int base_pix = NBDT_PIN(org.base_pix), last_pix = NBDT_PIN(org.last_pix);
#undef NBDT_PIN
  const int tap_w_v = lane < 9 ? d.w_tap[lane < 9 ? lane : 0] * cin : 0;   // lane t holds tap t's weight k-offset

  const int a_lane_pix = lane >> 2;
  const int a_lane_el = (((lane & 3) ^ ((lane >> 4) & 3)) << 3);
  int w_src[IPW_W];
#pragma unroll
  for (int k = 0; k < IPW_W; ++k) {
    const int id = wave + NWV * k;
    int row = id * 16 + (lane >> 2);
    row = row < BN ? row : BN - 1;
    w_src[k] = tiled ? n_blk * kchunks * 9 * (BN * 32) + id * 512 + lane * 8
                     : (n0 + row) * w_row_len + (((lane & 3) ^ ((row >> 2) & 3)) << 3);
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned w_ring = lds_base + 2 * a_bytes;

  auto issue_a = [&](int buf, int kc) {
    const unsigned dst0 = lds_base + buf * a_bytes;
#pragma unroll
    for (int k = 0; k < MAX_A_SLOTS; ++k)
      if (k < a_slots) {   // wave-uniform
        const int id = wave + NWV * k;
        int px = base_pix + id * 16 + a_lane_pix;
        px = px < last_pix ? px : last_pix;
        glds16(in_base + ((size_t)px * cin + kc * BK + a_lane_el), __builtin_amdgcn_readfirstlane(dst0 + id * 1024));
      }
  };
  auto issue_w = [&](int slot, int tap, int kc) {
    const int w_k = tiled ? (kc * 9 + tap) * (BN * 32) : __builtin_amdgcn_readlane(tap_w_v, tap) + kc * BK;
    const unsigned dst0 = w_ring + slot * W_BYTES;
#pragma unroll
    for (int k = 0; k < IPW_W; ++k) {
      const int id = wave + NWV * k;
      if (id < W_INSTR) glds16(w_base + (w_src[k] + w_k), __builtin_amdgcn_readfirstlane(dst0 + id * 1024));
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

  auto compute = [&](int abuf, int wslot, int tap) {
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

  // ---- pipeline: W tiles are prefetched PD = NWS-1 steps ahead through an NWS-slot ring
  constexpr int PD = NWS - 1;
  issue_a(0, 0);
#pragma unroll
  for (int i = 0; i < PD; ++i) issue_w(i, i, 0);     // nk >= 9 > PD: taps 0..PD-1 of slice 0
  int tap = 0, kc = 0, wslot = 0;
  int a_age = 1 << 20;   // steps since the last A halo tile was issued (it sits between two W tiles in issue order)
  for (int t = 0; t < nk; ++t) {
    // issued after W(t): W(t+1) .. W(min(t+PD-1, nk-1)), plus an A tile if one was issued in the last PD-1 steps
    int after = nk - 1 - t;
    after = after < PD - 1 ? after : PD - 1;
    int n = after * MINW;
    if (a_age <= PD - 1 && after > 0) n += a_min;
    wait_vm(n);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ++a_age;
    if (tap == 0 && kc + 1 < kchunks) {
      issue_a((kc + 1) & 1, kc + 1);
      a_age = 1;
    }
    if (t + PD < nk) {
      int t2 = tap + PD, k2 = kc;
      if (t2 >= 9) { t2 -= 9; ++k2; }
      int s2 = wslot + PD;
      s2 = s2 >= NWS ? s2 - NWS : s2;
      issue_w(s2, t2, k2);
    }
    compute(kc & 1, wslot, tap);
    wslot = wslot + 1 == NWS ? 0 : wslot + 1;
    if (++tap == 9) { tap = 0; ++kc; }
  }

  conv_epilogue<NT, HAS_RES, STATS, NWV>(acc, p, epi_lds_packed<NT, NWV>(smem, wave), m0, n0, m_blk, wave, lane, tid);
}

namespace nbdt {

// tickets of the split-K tiles: one zeroed array per (device, stream), like the det_rows workspace; the last block of a
// tile puts its ticket back to 0, so the array is all zero between launches
constexpr int kMaxSplitTiles = 4096;
static std::mutex g_ticket_mutex;
static std::map<std::pair<int, hipStream_t>, unsigned*> g_tickets;
static unsigned* split_tickets(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_ticket_mutex);
  unsigned*& t = g_tickets[std::make_pair(dev, st)];
  if (t) return t;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
  unsigned* q = nullptr;
  if (hipMalloc((void**)&q, kMaxSplitTiles * sizeof(unsigned)) != hipSuccess) return nullptr;
  if (hipMemset(q, 0, kMaxSplitTiles * sizeof(unsigned)) != hipSuccess) { (void)hipFree(q); return nullptr; }
  t = q;
  return t;
}

// Blocks per half tile the launch rule asks for (1 = no split); `items` = half tiles of the launch.  Pure host arithmetic
// (nbdt_conv_plan reports it; launch_halo applies it when the workspace and tickets exist).
int conv_ksplit_rule(const nbdt_conv_desc& d, int nt, int items) {
  const int cus = 8 * (32 - (reserved_cus() + 7) / 8);
  const int kchunks = d.cin / 32;
  int S = 1;
  if (d.ksplit > 1) S = std::min(d.ksplit, kchunks);
  else if (d.ksplit == 0 && d.wide_tile != 5 && 9 * kchunks >= 128) {
    const double want = std::sqrt((nt == 5 ? 0.45 : 0.33) * 9.0 * kchunks / 2.8);
    // wide_tile 1 = the caller says the launch has the GPU to itself (forward); 0 = a weight gradient runs beside it
    // (data gradients): there the split may fill half the CUs only -- in ResNet18 / 64x64 training steps 256 split
    // blocks beside the weight gradient cost 60 us per step more than 128 unsplit ones (profiles/r05_half_tile_ab.txt)
    const int room = d.wide_tile == 1 ? cus : cus / 2;
    S = std::min(std::min(room / std::max(items, 1), kchunks / 2), std::min(4, (int)(want + 0.5)));
  }
  if (items > kMaxSplitTiles) S = 1;
  return S < 1 ? 1 : S;
}

static int cout_tile(int cout) {
  const int nt32 = cout / 32;
  return nt32 % 5 == 0 ? 5 : (nt32 % 4 == 0 ? 4 : (nt32 % 2 == 0 ? 2 : 1));
}

// (cout tile, items) of a launch with this tile geometry: the ONE place the launcher and nbdt_conv_plan take them from
int conv_halo_items(const nbdt_conv_desc& d, const HaloGeom& hg, int M, int* nt_out) {
  const int nt = cout_tile(d.cout);
  const int bmh = 32 * hg.mw * hg.nwv;
  if (nt_out) *nt_out = nt;
  return ((M + bmh - 1) / bmh) * (d.cout / (32 * nt));
}

thread_local const char* g_last_igemm = "";     // nbdt_debug_last_igemm(): which kernel the last launch used
thread_local char g_last_igemm_full[128] = "";

// KIND 0: conv3x3_pp_kernel<.., 8> (ping-pong), 1: conv3x3_pp_kernel<.., 4> (same segments, one group, two blocks
// per CU), 2: conv3x3_halo_kernel (4 waves, plain weight layout), 3: conv3x3_pp_kernel<.., 8, false, 1> (ping-pong,
// 256-pixel half tiles)
template <int NT, int KIND, bool PAD>
static int launch_halo(ConvDmaParams& p, const HaloGeom& hg, hipStream_t st) {
  constexpr int NWV = (KIND == 0 || KIND == 3) ? 8 : 4;
  constexpr int MW = KIND == 3 ? 1 : 2;
  constexpr int BN = 32 * NT;
  constexpr int BMH = 32 * MW * NWV;
  p.n_blocks = p.d.cout / BN;
  p.m_blocks = (p.M + BMH - 1) / BMH;
  int items = conv_halo_items(p.d, hg, p.M, nullptr);      // (= m_blocks * n_blocks for this instantiation's NT, MW, NWV)
  if (items != p.m_blocks * p.n_blocks) return nbdt::fail(NBDT_EINVAL, "%s%s", "conv_halo_items disagrees with the launcher", "");
  p.ksplit = 1; p.k_ws = nullptr; p.k_tickets = nullptr;
  if (KIND == 3) {
    // Half tiles on at most half the CUs with a LONG K loop (ResNet18's last stage at batch 128: 32 tiles of 144 K steps):
    // split K.  A block's time is ~ b * steps / S for the loop plus ~2.8 us per split for the exchange (the tile's last
    // block reads every partial), b ~ 0.33 us per step (0.45 at 160 couts): S = sqrt(b * steps / 2.8), capped by the CUs,
    // by two slices per block and by 4; below 128 steps the split never paid (profiles/r05_half_tile_ab.txt).
    // desc.ksplit: 0 = this rule (automatic tile choice only), 1 = never, n = n blocks per tile (tests, A/B).  Needs the
    // per-stream workspace and tickets: without them (first use inside a hipGraph capture) the launch does not split.
    const int S = conv_ksplit_rule(p.d, NT, items);
    if (S >= 2) {
      float* ws = det_rows(st, (size_t)items * S * BN * 256);
      unsigned* tickets = ws ? split_tickets(st) : nullptr;
      if (ws && tickets) {
        p.ksplit = S; p.k_ws = ws; p.k_tickets = tickets;
        items *= S;
      }
    }
  }
  p.per_xcd = (items + 7) / 8;
  size_t shmem = 2 * (size_t)hg.a_bytes + (size_t)3 * BN * BK * 2;
  const size_t epi = conv_epilogue_lds_bytes<NT, NWV>();
  if (shmem < epi) shmem = epi;
  // The ping-pong kernel is persistent: one block per CU walks its XCD's items (the other two kinds keep one
  // block per item).  If the epilogue's LDS fits around halo buffer 0 and ring slots 0/1 -- regions in halo
  // buffer 1 and behind slot 1 -- a tile's first DMA is issued during the previous tile's epilogue.
  p.overlap = 0;
  int per_round = p.per_xcd;
  if (KIND == 0 || KIND == 3) {
    constexpr size_t REGION = 32 * (2 * BN + 16);
    const size_t in_a1 = std::min<size_t>(hg.a_bytes / REGION, NWV);
    const size_t need = 2 * (size_t)hg.a_bytes + 2 * (size_t)BN * BK * 2 + (NWV - in_a1) * REGION + NWV * 64 * 4 +
                        2 * BN * 4;
    if (need <= 160 * 1024) {
      p.overlap = 1;
      if (shmem < need) shmem = need;
    }
    // 32 CUs per XCD, minus the ones reserved for a collective's kernels (nbdt_set_reserved_cus, spread over the XCDs):
    // a persistent block that found its CU held by an RCCL block would start when that block ends, a millisecond late
    per_round = std::min(p.per_xcd, std::max(1, 32 - (reserved_cus() + 7) / 8));
#ifdef NBDT_PP_NO_PERSIST                    // timing experiment: one block per item, packed epilogue LDS
    per_round = p.per_xcd;
    p.overlap = 0;
#endif
  }
  static DeviceAttr site;     // one per (NT, KIND, PAD) instantiation; re-raised when a launch needs more LDS
  const dim3 grid(per_round * 8), blk(64 * NWV);
#define NBDT_KERNEL(R, S) \
  (KIND == 2 ? reinterpret_cast<const void*>(&conv3x3_halo_kernel<NT, R, S>) \
             : reinterpret_cast<const void*>(&conv3x3_pp_kernel<NT, R, S, NWV, PAD, MW>))
  if (site.need(shmem)) {
#define NBDT_ATTR(R, S) \
  NBDT_ATTR_CHECK(site, hipFuncSetAttribute(NBDT_KERNEL(R, S), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))
    NBDT_ATTR(true, 1); NBDT_ATTR(true, 0); NBDT_ATTR(false, 1); NBDT_ATTR(false, 0); NBDT_ATTR(false, 2);
    NBDT_ATTR(true, 3); NBDT_ATTR(false, 3);
#undef NBDT_ATTR
    site.done(shmem);
  }
  void* args[] = {(void*)&p, (void*)&hg};
#define NBDT_GO(R, S)                                                                                               \
  do {                                                                                                                 \
    if (KIND == 2) snprintf(g_last_igemm_full, sizeof(g_last_igemm_full), "conv3x3_halo_kernel<%d, %s, %d>", NT, R ? "true" : "false", S); \
    else snprintf(g_last_igemm_full, sizeof(g_last_igemm_full), "conv3x3_pp_kernel<%d, %s, %d, %d, %s, %d>", NT,     \
                  R ? "true" : "false", S, NWV, PAD ? "true" : "false", MW);                                        \
    NBDT_HIP_CHECK(hipLaunchKernel(NBDT_KERNEL(R, S), grid, blk, args, shmem, st));                                 \
  } while (0)
  if (p.aff_scale != nullptr) { if (p.res != nullptr) NBDT_GO(true, 3); else NBDT_GO(false, 3); }
  else if (p.bn_x != nullptr) NBDT_GO(false, 2);
  else if (p.res != nullptr) { if (p.stats) NBDT_GO(true, 1); else NBDT_GO(true, 0); }
  else { if (p.stats) NBDT_GO(false, 1); else NBDT_GO(false, 0); }
#undef NBDT_GO
#undef NBDT_KERNEL
  g_last_igemm = KIND == 0 ? (PAD ? "conv3x3_pp_kernel/pad" : "conv3x3_pp_kernel")
                           : KIND == 3 ? (p.ksplit > 1 ? "conv3x3_pp_kernel/half/ksplit" : "conv3x3_pp_kernel/half") : KIND == 1 ? "conv3x3_pp_kernel/4w" : "conv3x3_halo_kernel";
  return NBDT_OK;
}


// Fills hg when the pixel tiles of `tile` pixels are whole image rows / whole images and the two halo buffers
// fit in LDS next to the weight ring.  pad: the LDS image gets rows of gw + 4 slots (conv3x3_pp_kernel<.., PAD = true>).
static bool halo_geom_try(const nbdt_conv_desc* d, int tile, int nwv, bool pad, HaloGeom* hg) {
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
  hg->pad = pad ? 2 : 0;
  hg->dv = pad ? 4 : 0;
  hg->lpitch = hg->hw2 + hg->pad;
  hg->limg = (rb + 2) * hg->lpitch;
  hg->row_magic = (65536 + hg->lpitch - 1) / hg->lpitch;
  const int instr = (ib * hg->limg * 4 + 63) / 64;
  // pieces per wave: the 4-wave kernels keep up to 10 in flight per slice (conv3x3_halo_kernel) / have 2 x 7 issue slots
  // (conv3x3_pp_kernel<.., 4>); the ping-pong kernel issues one per wave per step at taps 0..6 (7 x 8 = 56 pieces), and
  // its padded form a second one for waves 0-7 at tap 0 (64)
  if (instr < nwv) return false;
  if ((instr + nwv - 1) / nwv > (nwv == 4 ? 10 : (pad ? 8 : 7))) return false;
  hg->a_instr = instr;
  hg->a_bytes = instr * 1024;
  hg->blocks_per_img = ib == 1 ? gh / rb : 1;
  hg->nwv = nwv;
  hg->mw = tile == 32 * nwv ? 1 : 2;
  const int lds = 2 * hg->a_bytes + 3 * cout_tile(d->cout) * 32 * BK * 2;
  return lds <= (nwv == 4 ? 80 : 156) * 1024;   // 4 waves: 2 blocks per CU (80 KiB each); 8 waves: 1 block per CU
}
// pad: the caller asked for the padded LDS pitch (desc.wide_tile = 4; 8-wave kernel, images narrower than 32 pixels).
// It is NOT what a launch gets by default: measured on MI355X (profiles/r05_lds_pitch_ab.txt) the padded kernel has no
// bank conflicts left and is 1-2 % SLOWER than the contiguous image (16x16x320: 171 vs 169 us, 8x8x640: 164 vs 161) --
// the conflicts sat in the load segment's slack under the partner's MFMA segment, while the extra address arithmetic of
// the padded form sits in the MFMA segment's shadow, which has none.
static bool halo_geom_for(const nbdt_conv_desc* d, int tile, int nwv, bool pad, HaloGeom* hg) {
  if (pad) return nwv == 8 && d->gw < 32 && d->gw % 4 == 0 && halo_geom_try(d, tile, nwv, true, hg);
  return halo_geom_try(d, tile, nwv, false, hg);
}

// Dense 3x3 / stride-1 conv over a padded NHWC tensor?  With DMA-ordered weight tiles: the ping-pong kernel, on 512-pixel
// tiles when they give at least 3/4 of the 256 CUs a block (every WRN-28-10 layer at 512 images per GPU), on 256-pixel
// half tiles when those fit the CUs in one round; without them the 4-wave 256-pixel kernels.
// desc.wide_tile: 0/1 automatic, 2 force 512-pixel tiles, 3 force the 4-wave 256-pixel kernel, 4 force 512-pixel tiles
// with the padded LDS pitch, 5 force half tiles (tests, A/B).
bool conv_halo_applicable(const nbdt_conv_desc* d, int M, HaloGeom* hg) {
#ifdef NBDT_HALO_NO_ACCUMULATE          // A/B builds: accumulating data gradients on the first-generation kernel (rounds 1-3)
  if (d->accumulate) return false;
#endif
  // (an accumulating launch passes its own output as the residual: every element is read and written by the one thread
  // that owns it, so in place is safe)
  if (d->ntaps != 9 || d->in_base != 0) return false;
  if (d->in_ws != d->cin || d->in_hs != (d->gw + 2) * d->cin || d->in_bs != (d->gh + 2) * d->in_hs) return false;
  for (int t = 0; t < 9; ++t)
    if (d->tap_off[t] != (t / 3) * d->in_hs + (t % 3) * d->in_ws) return false;
  // the halo kernels address the input with 32-bit byte offsets from the tensor base
  if ((long long)d->B * d->in_bs * 2 >= (1ll << 32)) return false;
  // the ping-pong kernel reads DMA-ordered weight tiles only (every engine launch has them; a caller without
  // them gets the 256-pixel kernel, which also reads the plain [cout][tap][cin] layout)
  bool tiled = d->w_tiled != 0 && d->w_ntaps == 9;
  for (int t = 0; t < 9; ++t) tiled = tiled && d->w_tap[t] == t;
  const int n_blocks = d->cout / (32 * cout_tile(d->cout));
  const long long tiles512 = ((long long)M + 511) / 512 * n_blocks;
  const long long tiles256 = ((long long)M + 255) / 256 * n_blocks;
  if (d->wide_tile == 4) return tiled && halo_geom_for(d, 512, 8, true, hg);     // padded LDS pitch or an error
  if (d->wide_tile == 5) return tiled && halo_geom_for(d, 256, 8, false, hg);    // half tiles or an error
  // Round 5: grids that give 512-pixel tiles to fewer than 3/4 of the CUs.  With at most one 256-pixel tile per CU the
  // ping-pong kernel on HALF tiles beats the 4-wave kernel by 10-30 % (one 4-wave block per CU has no partner to overlap
  // its load segment with: 149 -> 108 us at 256 x 8x8x640, profiles/r05_half_tile_ab.txt).  With more half tiles than CUs
  // (a collective holds some: nbdt_set_reserved_cus) one round of 512-pixel tiles is shorter than two rounds of half
  // tiles, and it never lost to the 4-wave kernel either (126 vs 149 us, same shape).
  const int cus = 8 * (32 - (reserved_cus() + 7) / 8);
  const bool pick = d->wide_tile != 2 && d->wide_tile != 3;
  const bool want_half = tiled && pick && tiles512 < 192 && tiles256 <= cus;
  if (want_half && halo_geom_for(d, 256, 8, false, hg)) return true;
  const bool want_wide = tiled && d->wide_tile != 3;
  if (want_wide && halo_geom_for(d, 512, 8, false, hg)) return true;
  if (d->wide_tile == 2) return false;
  // (the 4-wave form keeps the contiguous halo image at every width: its two pieces per wave and step with the padded
  //  addressing do not get through hipcc -- "s"-constrained asm operands come out as VGPRs -- and it only serves grids
  //  too small for 512-pixel tiles)
  return halo_geom_for(d, 256, 4, false, hg);
}

int conv3x3_halo(const nbdt_conv_desc* d, const HaloGeom& hg, const void* in, const void* w, void* out,
                 const void* res, float* stats, const BnBwdArgs* bn, int M, hipStream_t st) {
  ConvDmaParams p;
  p.d = *d;
  p.in = (const bf16_t*)in;
  p.w = (const bf16_t*)w;
  p.w_tiled = nullptr;
  if (d->w_tiled != 0) {   // only with the identity tap map (forward weights / already tap-reversed dgrad copy)
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
  p.deterministic = deterministic() ? 1 : 0;
  const int nt = cout_tile(d->cout);
#define NBDT_DISPATCH(KIND, PAD)                                     \
  {                                                                  \
    if (nt == 5) return launch_halo<5, KIND, PAD>(p, hg, st);        \
    if (nt == 4) return launch_halo<4, KIND, PAD>(p, hg, st);        \
    if (nt == 2) return launch_halo<2, KIND, PAD>(p, hg, st);        \
    return launch_halo<1, KIND, PAD>(p, hg, st);                     \
  }
  // (a padded LDS pitch is only ever chosen together with DMA-ordered weights, i.e. for conv3x3_pp_kernel)
  if (hg.pad != 0 && p.w_tiled == nullptr) return nbdt::fail(NBDT_EINVAL, "%s%s", "padded halo pitch without tiled weights", "");
  if (hg.mw == 1 && p.w_tiled == nullptr) return nbdt::fail(NBDT_EINVAL, "%s%s", "half tiles without tiled weights", "");
  if (hg.pad != 0 && hg.nwv != 8) return nbdt::fail(NBDT_EINVAL, "%s%s", "padded halo pitch is the 8-wave kernel's", "");
  if (hg.nwv == 8 && hg.mw == 1) NBDT_DISPATCH(3, false)
  if (hg.nwv == 8) { if (hg.pad) NBDT_DISPATCH(0, true) else NBDT_DISPATCH(0, false) }
  if (p.w_tiled != nullptr) NBDT_DISPATCH(1, false)
  NBDT_DISPATCH(2, false)
#undef NBDT_DISPATCH
}

}  // namespace nbdt
