// Shared pieces of the LDS-DMA implicit-GEMM kernels (conv_dma.hip, conv_halo.hip).
#pragma once
#include "common.h"
#include <stdlib.h>

using namespace nbdt;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
using nbdt::u32x4_t;

namespace nbdt {
struct ConvDmaParams {
  nbdt_conv_desc d;
  const bf16_t* in;
  const bf16_t* w;
  const bf16_t* w_tiled;   // nullable: DMA-ordered tiles (nbdt_weight_tile_batched), dense 3x3 kernel only
  bf16_t* out;
  const bf16_t* res;
  float* stats;        // nullable: [m_blocks][2][cout] per-pixel-tile partial sums (see STATS modes below)
  // STATS == 2 only: the BatchNorm whose input gradient this launch produces (x = its forward input)
  const bf16_t* bn_x;
  const float *bn_mean, *bn_rstd, *bn_gamma, *bn_beta;
  // STATS == 3 only (inference): out = act(conv * aff_scale[c] + aff_shift[c] [+ residual])
  const float *aff_scale, *aff_shift;
  int aff_act;         // 0 none, 1 ReLU, 2 swish
  int M, n_blocks, m_blocks, per_xcd;
  int overlap;         // conv3x3_pp_kernel: the epilogue's LDS is laid out around the next tile's first slice
  int deterministic;   // nbdt_set_deterministic: the block's statistics are summed wave by wave, not with LDS atomics
  // conv3x3_pp_kernel on half tiles, grids of at most half the CUs: `ksplit` blocks share an output tile, each takes a
  // contiguous range of the input-channel slices, writes its accumulators to k_ws ([tile][split][float4 q][thread]) and
  // takes a ticket; the block that draws the last one sums the ksplit partials in split order and runs the epilogue.
  int ksplit;          // >= 1
  float* k_ws;
  unsigned* k_tickets; // [tiles], zero between launches (the last block of a tile resets its ticket)
};
}  // namespace nbdt

constexpr int BM = 256;
constexpr int BK = 32;
constexpr int NSTAGE = 3;

__device__ __forceinline__ int lds_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ int pix_offset(int m, int gh, int gw, int bs, int hs, int ws, int base) {
  const int j = m % gw;
  const int t = m / gw;
  const int i = t % gh;
  const int b = t / gh;
  return b * bs + i * hs + j * ws + base;
}

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

// ------------------------------------------------------------------------------------------------
// Epilogue shared by the DMA kernels.  STATS modes (the per-tile partials go to p.stats):
//   0  none
//   1  forward : sum(out), sum(out^2)                  -> batch statistics of the NEXT BatchNorm
//   2  backward: out = dL/d(relu(bn(x))); with g' = out * [bn(x) > 0]:  sum(g'), sum(g' * xhat)
//                -> dbeta / dgamma sums of the BatchNorm being differentiated (replaces bn_bwd_reduce:
//                   the gradient tensor is not re-read, x is read once, coalesced, right here)
//   3  inference: eval-mode BatchNorm (running statistics folded into a per-channel scale/shift) and the
//                activation applied to the accumulators -- the BatchNorm/activation pass disappears
#ifdef NBDT_EPI_TIMING
#define NBDT_EPI_STAMP(i) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); epi_t[i] = (unsigned)t_; }
#else
#define NBDT_EPI_STAMP(i)
#endif
// Where the epilogue keeps its LDS: one transposition region per wave, the pixel row table, the block's statistics.
struct EpiLds {
  unsigned char* region;   // this wave's 32 rows x (2*BN + 16) bytes
  int* row_off;            // this wave's 64 entries
  float* blk_stats;        // [2][BN], shared by the block
  // where EVERY wave's region is (the statistics fold reads all of them): waves [0, n_a) at base_a + w * REGION,
  // the others at base_b + (w - n_a) * REGION
  unsigned char *base_a, *base_b;
  int n_a;
};
template <int NT, int NWV>
__device__ __forceinline__ EpiLds epi_lds_packed(unsigned char* smem, int wave) {   // everything from smem + 0
  constexpr int REGION = 32 * (2 * 32 * NT + 16);
  EpiLds l;
  l.region = smem + wave * REGION;
  l.row_off = (int*)(smem + NWV * REGION) + wave * 64;
  l.blk_stats = (float*)(smem + NWV * REGION + NWV * 64 * 4);
  l.base_a = smem; l.base_b = smem; l.n_a = NWV;
  return l;
}
struct EpiNoHook { __device__ __forceinline__ void operator()() const {} };

// `after_entry` runs once every wave of the block has passed the entry barrier (nobody reads the K ring any
// more) and before the first LDS write: the persistent kernel issues the next tile's first LDS-DMA there.
template <int NT, bool HAS_RES, int STATS, int NWV = 4, int MW = 2, typename HOOK = EpiNoHook>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[NT][MW], const nbdt::ConvDmaParams& p,
                                              const EpiLds& lds, int m0, int n0, int m_blk, int wave, int lane,
                                              int tid, unsigned* epi_t = nullptr, HOOK after_entry = HOOK()) {
  NBDT_EPI_STAMP(0)
  constexpr int BN = 32 * NT;
  constexpr int NTHR = 64 * NWV;
  const nbdt_conv_desc& d = p.d;
  const int frag_row = lane & 31;
  const int frag_half = lane >> 5;
  // ---- epilogue through LDS.  The accumulator layout (lane = pixel, regs = 4 consecutive couts) would
  // give 8-byte stores / residual loads scattered over 32 pixel rows per instruction (the v1 residual
  // epilogue cost 84 us per launch).  Instead each wave transposes its 32-pixel x BN tile through a
  // private LDS region: rows of PITCH = 2*BN + 16 bytes (16-B aligned, 2-way at worst for the 8-byte
  // lane writes), then walks it with a FIXED 8-channel chunk per lane: 16-byte coalesced residual
  // loads / output stores (20 lanes = one 320-B pixel row), and -- because the chunk is fixed -- the
  // per-channel sum and sum of squares of the bf16 output fall out of the same pass in registers.  They
  // feed the next BatchNorm (bn_finalize only), replacing a full re-read of the tensor (bn_stats_kernel).
  constexpr int PITCH = 2 * BN + 16;   // a wave's region is 32 rows of it (EpiLds::region)
  constexpr int NCH = BN / 8;            // 8-channel chunks per row
  constexpr int RL = 64 / NCH;           // row lanes: lanes [0, RL*NCH) are active in the row walk
  constexpr int ROW_ITERS = (32 + RL - 1) / RL;
  constexpr int REGION_BYTES = 32 * PITCH;
  // The block's statistics: every walker lane ends up with 16 partial sums (8 channels x {sum, second sum}).  They used
  // to go into blk_stats with LDS atomics (ds_add_f32, 16 per lane): 24 lanes of the block add to every address, and
  // that tail took 25 k cycles per 512-pixel tile -- 40-80 us of a 32x32x160 launch (s_memtime stamps,
  // profiles/r04_pp_epilogue_phases.txt).  Now each wave parks its lanes' partials in its own transposition region
  // (free after its last row walk), one barrier, and 2*BN threads add the NWV * RL partials of their channel in a fixed
  // order: no atomics, ~1 k cycles, and run-to-run identical bits without a deterministic-mode special case.
  // (NT == 1: a wave's 64 x 64 B of partials do not fit its 2.5 KB region -- those small launches keep the atomics.)
#ifdef NBDT_EPI_STATS_ATOMICS      // A/B build (scratch/variants): the LDS-atomic form everywhere
  constexpr bool STATS_VIA_REGIONS = false;
#else
  constexpr bool STATS_VIA_REGIONS = RL * NCH * 64 <= REGION_BYTES;
#endif
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();          // every wave is done with the K ring
  asm volatile("" ::: "memory");
  after_entry();
  unsigned char* region = lds.region;
  int* row_off = lds.row_off;              // element offset of each of the wave's 64 pixels
  float* blk_stats = lds.blk_stats;        // [2][BN]
  {
    const int m = m0 + wave * (32 * MW) + lane;     // (MW == 1: entries 32..63 belong to the next wave's pixels, never read)
    row_off[lane] = m < p.M ? pix_offset(m, d.gh, d.gw, d.out_bs, d.out_hs, d.out_ws, d.out_base) + n0 : -1;
    if ((STATS == 1 || STATS == 2) && !STATS_VIA_REGIONS) {
      for (int i = tid; i < 2 * BN; i += NTHR) blk_stats[i] = 0.f;
      __syncthreads();   // block-uniform: zeroed before any wave's atomics
    }
  }
  NBDT_EPI_STAMP(1)
  const int ch = lane % NCH, rl = lane / NCH;
  const bool walker = rl < RL;
  float s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
  float bmu[8], brs[8], bsc[8], bsh[8];   // STATS == 2: this lane's 8 channels of the BatchNorm
  if (STATS == 2 && walker) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = n0 + ch * 8 + i;
      bmu[i] = p.bn_mean[c];
      brs[i] = p.bn_rstd[c];
      bsc[i] = p.bn_gamma[c] * brs[i];
      bsh[i] = p.bn_beta[c] - bmu[i] * bsc[i];
    }
  }

#pragma unroll
  for (int tm = 0; tm < MW; ++tm) {
    // Row offsets of this lane's ROW_ITERS rows, fetched from LDS in ONE batch.  (The first version read
    // row_off[] inside every loop below: each iteration was  ds_read -> wait -> [ds_read_b128 -> wait] -> store,
    // two LDS round trips per row, 12 rows, twice: 8 k cycles of a 72 k-cycle tile, measured with s_memtime.)
    int offs[ROW_ITERS];
    bool live[ROW_ITERS];
#pragma unroll
    for (int it = 0; it < ROW_ITERS; ++it) {
      int r = rl + it * RL;
      r = r < 32 ? r : 31;
      offs[it] = walker ? row_off[tm * 32 + r] : -1;
    }
#pragma unroll
    for (int it = 0; it < ROW_ITERS; ++it) {
      live[it] = rl + it * RL < 32 && offs[it] >= 0;
      offs[it] = offs[it] >= 0 ? offs[it] : p.d.out_base + n0;     // rows past M: harmless in-bounds address
    }
    if (HAS_RES) {  // (1) residual rows -> LDS, coalesced; all loads issued before the first LDS write
      if (walker) {
        u32x4_t rv[ROW_ITERS];
#pragma unroll
        for (int it = 0; it < ROW_ITERS; ++it) rv[it] = *(const u32x4_t*)(p.res + offs[it] + ch * 8);
#pragma unroll
        for (int it = 0; it < ROW_ITERS; ++it) {
          const int r = rl + it * RL;
          if (r < 32) *(u32x4_t*)(region + r * PITCH + ch * 16) = rv[it];
        }
      }
    }
    // (2) accumulators (+ residual, fp32, single rounding) -> bf16 -> LDS at [pixel][cout]
    unsigned char* myrow = region + frag_row * PITCH + frag_half * 8;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v0 = acc[tn][tm][4 * q + 0], v1 = acc[tn][tm][4 * q + 1];
        float v2 = acc[tn][tm][4 * q + 2], v3 = acc[tn][tm][4 * q + 3];
        unsigned char* at = myrow + (tn * 32 + q * 8) * 2;
        if (STATS == 3) {   // this lane's 4 consecutive couts
          const int c = n0 + tn * 32 + q * 8 + frag_half * 4;
          const float4 sc = *(const float4*)(p.aff_scale + c);
          const float4 sh = *(const float4*)(p.aff_shift + c);
          v0 = v0 * sc.x + sh.x; v1 = v1 * sc.y + sh.y; v2 = v2 * sc.z + sh.z; v3 = v3 * sc.w + sh.w;
        }
        if (HAS_RES) {
          const u32x2 r = *(const u32x2*)at;
          v0 += __uint_as_float(r[0] << 16);
          v1 += __uint_as_float(r[0] & 0xffff0000u);
          v2 += __uint_as_float(r[1] << 16);
          v3 += __uint_as_float(r[1] & 0xffff0000u);
        }
        if (STATS == 3) {
          if (p.aff_act == 1) {
            v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;
          } else if (p.aff_act == 2) {
            v0 = v0 / (1.f + __expf(-v0)); v1 = v1 / (1.f + __expf(-v1));
            v2 = v2 / (1.f + __expf(-v2)); v3 = v3 / (1.f + __expf(-v3));
          }
        }
        u32x2 pk;
        pk[0] = pack_bf16x2(v0, v1);
        pk[1] = pack_bf16x2(v2, v3);
        *(u32x2*)at = pk;
      }
    NBDT_EPI_STAMP(2 + 2 * tm)
    // (3) walk the rows: coalesced 16-byte stores (+ statistics of the rounded values).  Three batches -- BN-input
    // loads, LDS row reads, stores -- so every load of a batch is in flight before the first wait.
    if (walker) {
      u32x4_t xv[STATS == 2 ? ROW_ITERS : 1];
      if (STATS == 2) {
#pragma unroll
        for (int it = 0; it < ROW_ITERS; ++it) xv[it] = *(const u32x4_t*)(p.bn_x + offs[it] + ch * 8);
      }
      u32x4_t ov[ROW_ITERS];
#pragma unroll
      for (int it = 0; it < ROW_ITERS; ++it) {
        int r = rl + it * RL;
        r = r < 32 ? r : 31;
        ov[it] = *(const u32x4_t*)(region + r * PITCH + ch * 16);
      }
#pragma unroll
      for (int it = 0; it < ROW_ITERS; ++it)
        if (live[it]) st16(p.out + offs[it] + ch * 8, ov[it]);
      if (STATS == 1) {
#pragma unroll
        for (int it = 0; it < ROW_ITERS; ++it) {
          float f[8];
          unpack8(ov[it], f);
          const float m = live[it] ? 1.f : 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float v = f[i] * m; s1[i] += v; s2[i] += v * v; }
        }
      }
      if (STATS == 2) {
#pragma unroll
        for (int it = 0; it < ROW_ITERS; ++it) {
          float f[8], fx[8];
          unpack8(ov[it], f);
          unpack8(xv[it], fx);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float gg = (live[it] && (fx[i] * bsc[i] + bsh[i]) > 0.f) ? f[i] : 0.f;
            s1[i] += gg;
            s2[i] += gg * ((fx[i] - bmu[i]) * brs[i]);
          }
        }
      }
    }
    NBDT_EPI_STAMP(3 + 2 * tm)
  }
  if (STATS == 1 || STATS == 2) {
    constexpr int RPB = NWV * MW / 8;      // 256-pixel rows of statistics per tile
    // one partial row per pixel tile, plain stores (no global atomics: 2048 blocks x 320 atomics cost more than the
    // separate statistics pass they replace); nbdt_bn_finalize folds the rows
    // (the fold kernels expect one row per 256 pixels: a 512-pixel tile fills row 2*m_blk and zeroes the next)
    float* part = p.stats + (size_t)m_blk * RPB * 2 * d.cout;
    const bool second = RPB == 2 && (m_blk * RPB + 1) * 256 < p.M;
    if (STATS_VIA_REGIONS) {
      if (walker) {      // this lane's 16 partials -> [rl][ch][16 floats] at the start of the wave's own region
        float* mine = (float*)region + (rl * NCH + ch) * 16;
        *(float4*)(mine + 0) = make_float4(s1[0], s1[1], s1[2], s1[3]);
        *(float4*)(mine + 4) = make_float4(s1[4], s1[5], s1[6], s1[7]);
        *(float4*)(mine + 8) = make_float4(s2[0], s2[1], s2[2], s2[3]);
        *(float4*)(mine + 12) = make_float4(s2[4], s2[5], s2[6], s2[7]);
      }
      __syncthreads();   // (also: every wave's row walk has read its region before anybody's partials land in it --
                         //  a wave only writes its OWN region, after its own walk)
      for (int i = tid; i < 2 * BN; i += NTHR) {
        const int which = i / BN, c = i - which * BN;
        const int off = ((c >> 3) * 16 + which * 8 + (c & 7)) * 4;      // byte offset inside an [rl] slab
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
          const unsigned char* rg = w < lds.n_a ? lds.base_a + w * REGION_BYTES : lds.base_b + (w - lds.n_a) * REGION_BYTES;
#pragma unroll
          for (int r = 0; r < RL; ++r) sum += *(const float*)(rg + r * (NCH * 64) + off);
        }
        part[(size_t)which * d.cout + n0 + c] = sum;
        if (second) part[(size_t)(2 + which) * d.cout + n0 + c] = 0.f;
      }
    } else {
      if (p.deterministic) {
        // fixed order: wave 0's row lanes 0, 1, ..., then wave 1's, ... -- one turn per barrier, plain read-modify-write
        // (lanes of one turn hold distinct channel chunks).  NWV * RL barriers per tile: a debugging mode, not a fast one.
        for (int w = 0; w < NWV; ++w)
          for (int r = 0; r < RL; ++r) {
            if (wave == w && walker && rl == r)
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                blk_stats[ch * 8 + i] += s1[i];
                blk_stats[BN + ch * 8 + i] += s2[i];
              }
            __syncthreads();
          }
      } else {
        if (walker)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            atomicAdd(blk_stats + ch * 8 + i, s1[i]);
            atomicAdd(blk_stats + BN + ch * 8 + i, s2[i]);
          }
        __syncthreads();
      }
      for (int i = tid; i < 2 * BN; i += NTHR) {
        const int which = i / BN, c = i - which * BN;
        part[(size_t)which * d.cout + n0 + c] = blk_stats[i];
        if (second) part[(size_t)(2 + which) * d.cout + n0 + c] = 0.f;
      }
    }
  }
}

template <int NT, int NWV = 4>
constexpr int conv_epilogue_lds_bytes() {
  return NWV * 32 * (2 * 32 * NT + 16) + NWV * 64 * 4 + 2 * 32 * NT * 4;
}
