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

constexpr int NW_SLOTS = 3;

constexpr int min_w_dma_h(int w_instr) {
  int best = 1 << 30;
  for (int w = 0; w < 4; ++w) {
    int n = 0;
    for (int id = w; id < w_instr; id += 4) ++n;
    best = n < best ? n : best;
  }
  return best;
}


template <int NT, bool HAS_RES, int STATS>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(nbdt::ConvDmaParams p, nbdt::HaloGeom hg) {
  constexpr int BN = 32 * NT;
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int W_INSTR = W_BYTES / 1024;
  constexpr int IPW_W = (W_INSTR + 3) / 4;
  constexpr int MINW = min_w_dma_h(W_INSTR);
  constexpr int MAX_A_SLOTS = 10;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [A buf 0][A buf 1][W ring x3]

  const int bid = blockIdx.x;
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
  const int cin = NBDT_PIN(d.cin);
  const int kchunks = cin >> 5;
  const int nk = 9 * kchunks;
  const int w_row_len = d.w_ntaps * cin;
  const int a_bytes = NBDT_PIN(hg.a_bytes);
  const int a_instr = NBDT_PIN(hg.a_instr);
  const int a_slots = (a_instr - wave + 3) >> 2;     // DMA instructions this wave issues per halo tile
  const int a_min = a_instr >> 2;                    // fewest any wave issues (for the counted waits)
  const int hw2 = NBDT_PIN(hg.hw2), himg = NBDT_PIN(hg.himg), hp_total = NBDT_PIN(hg.hp);
  const unsigned long long in_u = (unsigned long long)p.in, w_u = (unsigned long long)p.w;
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

  // ---- A halo DMA slots: wave-instruction id = wave + 4k covers halo pixels [16*id, 16*id+16)
  const int cpos = lane & 3;
  int a_src[MAX_A_SLOTS];
#pragma unroll
  for (int k = 0; k < MAX_A_SLOTS; ++k) {
    int hp = (wave + 4 * k) * 16 + (lane >> 2);
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
    const int id = wave + 4 * k;
    int row = id * 16 + (lane >> 2);
    row = row < BN ? row : BN - 1;
    w_src[k] = (n0 + row) * w_row_len + ((cpos ^ ((row >> 2) & 3)) << 3);
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned w_ring = lds_base + 2 * a_bytes;

  auto issue_a = [&](int buf, int kc) {
    const unsigned dst0 = lds_base + buf * a_bytes;
#pragma unroll
    for (int k = 0; k < MAX_A_SLOTS; ++k)
      if (k < a_slots)   // wave-uniform
        glds16(in_base + (a_src[k] + kc * BK), __builtin_amdgcn_readfirstlane(dst0 + (wave + 4 * k) * 1024));
  };
  auto issue_w = [&](int slot, int tap, int kc) {
    const int w_k = __builtin_amdgcn_readlane(tap_w_v, tap) + kc * BK;
    const unsigned dst0 = w_ring + slot * W_BYTES;
#pragma unroll
    for (int k = 0; k < IPW_W; ++k) {
      const int id = wave + 4 * k;
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
  // halo index of this lane's two output pixels at tap (0,0)
  int hp0[2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int pl = wave * 64 + tm * 32 + frag_row;     // pixel inside the 256-pixel tile
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

  // counted wait: everything issued after W(t) may stay in flight
  auto wait_w = [&](bool a_after) {
    if (!a_after) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(MINW) : "memory");
    } else {
      switch (a_min) {
#define NBDT_CASE(K) case K: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(MINW + K) : "memory"); break;
        NBDT_CASE(1) NBDT_CASE(2) NBDT_CASE(3) NBDT_CASE(4) NBDT_CASE(5) NBDT_CASE(6) NBDT_CASE(7) NBDT_CASE(8)
        NBDT_CASE(9) NBDT_CASE(10)
#undef NBDT_CASE
        default: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(MINW) : "memory"); break;
      }
    }
  };

  // ---- pipeline
  issue_a(0, 0);
  issue_w(0, 0, 0);
  if (nk > 1) issue_w(1, 1, 0);
  int tap = 0, kc = 0, wslot = 0;
  bool a_issued_prev = false;   // did the previous step issue an A halo tile (after W(t), before W(t+1))?
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) wait_w(a_issued_prev);
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    a_issued_prev = false;
    if (tap == 0 && kc + 1 < kchunks) {
      issue_a((kc + 1) & 1, kc + 1);
      a_issued_prev = true;
    }
    if (t + 2 < nk) {
      int t2 = tap + 2, k2 = kc;
      if (t2 >= 9) { t2 -= 9; ++k2; }
      int s2 = wslot + 2;
      s2 = s2 >= NW_SLOTS ? s2 - NW_SLOTS : s2;
      issue_w(s2, t2, k2);
    }
    compute(kc & 1, wslot, tap);
    wslot = wslot + 1 == NW_SLOTS ? 0 : wslot + 1;
    if (++tap == 9) { tap = 0; ++kc; }
  }

  conv_epilogue<NT, HAS_RES, STATS>(acc, p, smem, m0, n0, m_blk, wave, lane, tid);
}

namespace nbdt {

template <int NT>
static int launch_halo(ConvDmaParams& p, const HaloGeom& hg, hipStream_t st) {
  constexpr int BN = 32 * NT;
  p.n_blocks = p.d.cout / BN;
  p.m_blocks = (p.M + BM - 1) / BM;
  const int items = p.m_blocks * p.n_blocks;
  p.per_xcd = (items + 7) / 8;
  size_t shmem = 2 * (size_t)hg.a_bytes + (size_t)NW_SLOTS * BN * BK * 2;
  const size_t epi = conv_epilogue_lds_bytes<NT>();
  if (shmem < epi) shmem = epi;
  static size_t attr_bytes = 0;
  if (shmem > attr_bytes) {
#define NBDT_ATTR(R, S)                                                                                     \
  NBDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<NT, R, S>),         \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))
    NBDT_ATTR(true, 1); NBDT_ATTR(true, 0); NBDT_ATTR(false, 1); NBDT_ATTR(false, 0); NBDT_ATTR(false, 2);
#undef NBDT_ATTR
    attr_bytes = shmem;
  }
  const dim3 grid(p.per_xcd * 8), blk(256);
#define NBDT_GO(R, S) hipLaunchKernelGGL((conv3x3_halo_kernel<NT, R, S>), grid, blk, shmem, st, p, hg)
  if (p.bn_x != nullptr) NBDT_GO(false, 2);
  else if (p.res != nullptr) { if (p.stats) NBDT_GO(true, 1); else NBDT_GO(true, 0); }
  else { if (p.stats) NBDT_GO(false, 1); else NBDT_GO(false, 0); }
#undef NBDT_GO
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

// Returns true (and fills hg) when the descriptor is a dense 3x3 / stride-1 conv whose 256-pixel tiles
// are whole image rows or whole images, and the halo tile fits in LDS next to the weight ring.
bool conv_halo_applicable(const nbdt_conv_desc* d, int M, HaloGeom* hg) {
  if (d->ntaps != 9 || d->in_base != 0 || d->accumulate) return false;
  if (d->in_ws != d->cin || d->in_hs != (d->gw + 2) * d->cin || d->in_bs != (d->gh + 2) * d->in_hs) return false;
  for (int t = 0; t < 9; ++t)
    if (d->tap_off[t] != (t / 3) * d->in_hs + (t % 3) * d->in_ws) return false;
  const int gw = d->gw, gh = d->gh;
  if (gw > 256 || 256 % gw != 0) return false;
  int ib, rb;
  if (gw * gh >= 256) {
    rb = 256 / gw;
    if (gh % rb != 0) return false;
    ib = 1;
  } else {
    if (256 % (gw * gh) != 0) return false;
    ib = 256 / (gw * gh);
    rb = gh;
  }
  hg->ib = ib; hg->rb = rb;
  hg->hw2 = gw + 2;
  hg->himg = (rb + 2) * (gw + 2);
  hg->hp = ib * hg->himg;
  const int instr = (hg->hp * 4 + 63) / 64;
  if ((instr + 3) / 4 > 10 || instr < 4) return false;
  hg->a_instr = instr;
  hg->a_bytes = instr * 1024;
  hg->blocks_per_img = ib == 1 ? gh / rb : 1;
  // 2 blocks per CU: 2 A buffers + weight ring must stay under 80 KiB
  const int nt32 = d->cout / 32;
  const int nt = nt32 % 5 == 0 ? 5 : (nt32 % 4 == 0 ? 4 : (nt32 % 2 == 0 ? 2 : 1));
  if (2 * hg->a_bytes + NW_SLOTS * nt * 32 * BK * 2 > 80 * 1024) return false;
  (void)M;
  return true;
}

int conv3x3_halo(const nbdt_conv_desc* d, const HaloGeom& hg, const void* in, const void* w, void* out,
                 const void* res, float* stats, const BnBwdArgs* bn, int M, hipStream_t st) {
  ConvDmaParams p;
  p.d = *d;
  p.in = (const bf16_t*)in;
  p.w = (const bf16_t*)w;
  p.out = (bf16_t*)out;
  p.res = (const bf16_t*)res;
  p.stats = stats;
  p.bn_x = bn ? (const bf16_t*)bn->x : nullptr;
  p.bn_mean = bn ? bn->mean : nullptr; p.bn_rstd = bn ? bn->rstd : nullptr;
  p.bn_gamma = bn ? bn->gamma : nullptr; p.bn_beta = bn ? bn->beta : nullptr;
  p.M = M;
  p.debug = 0;
  const int nt32 = d->cout / 32;
  if (nt32 % 5 == 0) return launch_halo<5>(p, hg, st);
  if (nt32 % 4 == 0) return launch_halo<4>(p, hg, st);
  if (nt32 % 2 == 0) return launch_halo<2>(p, hg, st);
  return launch_halo<1>(p, hg, st);
}

}  // namespace nbdt
