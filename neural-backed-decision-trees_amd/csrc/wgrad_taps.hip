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

using namespace nbdt;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace nbdt {
struct WgradTapsParams {
  nbdt_wgrad_desc d;
  const bf16_t* x;
  const bf16_t* gy;
  float* dw;
  int stages;            // M / 32
  int stages_per_split;
  int n_ci_blocks;       // cin / 32
  int splits, items, per_xcd;
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
        atomicAdd(p.dw + ((int64_t)co * d.w_ntaps + w_tap) * d.cin + ci, acc[t][a][r]);
      }
  }
}

namespace nbdt {

bool wgrad_taps_applicable(const nbdt_wgrad_desc* d) {
  if (d->ntaps != 9 || d->x_base != 0) return false;
  if (d->x_ws != d->cin || d->x_hs != (d->gw + 2) * d->cin || d->x_bs != (d->gh + 2) * d->x_hs) return false;
  for (int t = 0; t < 9; ++t)
    if (d->tap_off[t] != (t / 3) * d->x_hs + (t % 3) * d->x_ws) return false;
  const int gw = d->gw, gh = d->gh;
  if (!(gw % 32 == 0 || 32 % gw == 0)) return false;
  const int rs = gw >= 32 ? 1 : 32 / gw;
  if (gh % rs != 0) return false;
  return true;
}

template <int WM, int NWV>
static int launch_taps(WgradTapsParams& p, hipStream_t st) {
  const nbdt_wgrad_desc& d = p.d;
  p.n_ci_blocks = d.cin / (8 * NWV);
  const int tiles = (d.cout / (32 * WM)) * p.n_ci_blocks;
  int splits = (NWV == 4 ? 512 : 256) / tiles;
  const int max_splits = p.stages / 16 > 0 ? p.stages / 16 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.stages_per_split = (p.stages + splits - 1) / splits;
  splits = (p.stages + p.stages_per_split - 1) / p.stages_per_split;
  p.splits = splits;
  p.items = tiles * splits;
  p.per_xcd = (p.items + 7) / 8;
  const size_t shmem = (size_t)NSTAGE * ((32 * WM / 8) * 512 + x_bytes(NWV));
  static bool attr_set = false;
  if (!attr_set) {
    NBDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_taps_kernel<WM, NWV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad_taps_kernel<WM, NWV>), dim3(p.per_xcd * 8), dim3(64 * NWV), shmem, st, p);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

int wgrad_taps(const nbdt_wgrad_desc* d, const void* x, const void* gy, float* dw, hipStream_t st) {
  WgradTapsParams p;
  p.d = *d;
  p.x = (const bf16_t*)x;
  p.gy = (const bf16_t*)gy;
  p.dw = dw;
  const int M = d->B * d->gh * d->gw;
  p.stages = M / KS;
  p.rs = d->gw >= 32 ? 1 : 32 / d->gw;
  p.cs = d->gw >= 32 ? 32 : d->gw;
  p.hw2 = p.cs + 2;
  p.hp = (p.rs + 2) * (p.cs + 2);
  p.stages_per_row = d->gw / p.cs;
  p.rowgroups = d->gh / p.rs;
  p.div_spr = make_fastdiv((unsigned)p.stages_per_row);
  p.div_rg = make_fastdiv((unsigned)p.rowgroups);
  const int mt = d->cout / 32;
  if (mt % 5 == 0) return launch_taps<5, 4>(p, st);
  if (mt % 4 == 0) return launch_taps<4, 4>(p, st);
  if (mt % 2 == 0) return launch_taps<2, 4>(p, st);
  return launch_taps<1, 4>(p, st);
}

}  // namespace nbdt
