// "Slice list" implicit GEMM (gfx950 only): the shape-changing convolutions of a (Wide)ResNet as sums of stride-1
// tap-subset convolutions over tensors that share one padded pixel grid (include/nbdt_hip.h, nbdt_conv_seg_*).
//
// Replaces, for the units where the shape changes, the ops behind nbdt/models/resnet.py:56-67 (strided conv1 + 1x1
// shortcut) and pytorchcv's PreResUnit with stride 2 / a channel change (nbdt/models/wideresnet.py:1-5), forward and
// data gradient.  Rounds 1-5 ran them on conv_igemm_dma_kernel -- one LDS-DMA tile per tap, 4 lock-step waves, 0.14-0.33
// of the MFMA peak, 14 % of a WRN-28-10 step for 5 % of its flops.
//
// The structure is conv3x3_pp_kernel's (conv_halo.hip): 8 waves, the two waves of a SIMD alternate between a load
// segment (operand fragments LDS -> registers, LDS-DMA issue) and a pure-MFMA segment one barrier apart; per 32-channel
// K slice the block holds its pixel tile INCLUDING the one-pixel halo in LDS and every tap reads a shifted window of
// it; weights stream through a 3-slot ring of DMA-ordered tiles; persistent blocks.  What is new is that nothing about
// the K loop is a literal any more:
//   * a K step is described by a 32-byte record (SegStep) the wave fetches with one s_load one step ahead: which halo
//     buffer and tap offset it reads, and which LDS-DMA pieces of WHICH slice (tensor, channel, piece range) it issues;
//   * slices have 1..9 taps, so a slice's pieces cannot always go out "during the previous slice": the host schedules
//     them (seg_schedule: a small DP that spreads every slice's pieces over the steps between "its buffer is free" and
//     "two steps before its first read", minimising the squared pieces per step), with 2 or 3 halo buffers;
//   * a launch has up to 4 CLASSES with their own slice lists and output maps (the parity classes of a strided data
//     gradient); a persistent block walks the classes of its tiles one after the other, so every CU gets the same mix.
// Ordering rules (the only things that order LDS-DMA against ds_read are vmcnt + a barrier; same analysis as
// conv_halo.hip, with t = K step, group 0 = waves 0-3 one barrier ahead of group 1):
//   W(t+2) is issued in L(t) into the ring slot of W(t-1), waited for by its issuer in L(t+1), first read in L(t+2);
//   a halo piece issued in L(t) comes after that step's weight pieces; the wait of L(t+1) leaves it in flight (vmcnt
//   counts in order) unless the step is marked strict, the wait of L(t+2) retires it: readable from step t+3 (t+2 if
//   strict).  The schedule never issues a piece of slice s before the last step of slice s - nbuf (its buffer's previous
//   tenant) has been read by both groups, i.e. not before step last(s - nbuf) + 1.
#include "conv_common.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

namespace nbdt {

struct SegStep {          // one K step; 32 bytes, fetched with one s_load_dwordx8
  int toff;               // halo-pixel offset of the step's tap: R * (gw + 2) + S
  int abuf;               // byte offset of the halo buffer the step reads
  int pf_n;               // LDS-DMA rounds issued in this step's load segment: pieces id0 + 8 j + wave, j < pf_n
  int pf_dst;             // LDS byte offset of piece id0 (buffer + id0 * 1024)
  int pf_pix;             // halo pixel of piece id0: id0 * 16
  int pf_choff;           // first channel of the slice being fetched
  int pf_tensor;          // input tensor of the slice being fetched
  int strict;             // bit 0: the pieces must be readable two steps later (the next step waits vmcnt(0));
                          // bits 8..: waves that issue the step's last round
};
struct SegWStep { int matrix, w_off; };     // where a step's weight tile comes from (tiler, fp32 twin)
struct SegRefStep { int tensor, ch0, dy, dx, matrix, w_off; };

struct SegClassDev {
  int nsteps, npro, step0, pad0;
  long long w_off;        // element offset of the class's tiles in the DMA-ordered weight buffer
  int out_bs, out_hs, out_ws, out_base;
  int pro_tensor[2], pro_choff[2];     // slices loaded in the tile's prologue (buffers 0 .. npro-1)
};
struct ConvSegParams {
  ConvDmaParams c;        // what the shared epilogue reads (d.out_* are replaced per class)
  const bf16_t* in[4];
  int pix_stride[4];
  const SegStep* steps;
  const bf16_t* w_tiles;
  SegClassDev cls[4];
  int ncls, nbuf, tiles, tiles_per_xcd;
  int overlap;            // the epilogue's LDS is laid out around halo buffer 0 and ring slot 0 [and 1: early_w1]
  int early_w1;
};

}  // namespace nbdt

#ifndef NBDT_SEG_TIMING
#define NBDT_SEG_TIMING 0      // 1: s_memtime stamps around the segments of every K step, per-(item, wave) sums in g_seg_timing
#endif
#if NBDT_SEG_TIMING
__device__ unsigned g_seg_timing[8192 * 8];    // [item * 8 + wave][8]: load segment, barrier 1, MFMA segment, barrier 2, loop, steps, prologue, epilogue
extern "C" int nbdt_debug_seg_timing(unsigned* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_seg_timing), sizeof(unsigned) * 8192 * 8);
}
__device__ __forceinline__ unsigned seg_stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return (unsigned)t;
}
#define NBDT_SEG_STAMP(acc_) { const unsigned t_ = seg_stamp(); acc_ += t_ - tm_prev; tm_prev = t_; }
#else
#define NBDT_SEG_STAMP(acc_)
#endif

__device__ __forceinline__ void seg_wait_vm(int n) {     // the n most recent LDS-DMA of this wave may stay in flight
  switch (n) {
#define NBDT_CASE(K) case K: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
    NBDT_CASE(1) NBDT_CASE(2) NBDT_CASE(3) NBDT_CASE(4) NBDT_CASE(5) NBDT_CASE(6) NBDT_CASE(7)
#undef NBDT_CASE
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <typename T>
__device__ __forceinline__ T seg_sel4(int i, T a0, T a1, T a2, T a3) {
  return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
}

template <int NT, bool HAS_RES, int STATS, int MW>
__global__ __launch_bounds__(512, 2) void conv_seg_kernel(nbdt::ConvSegParams p, nbdt::HaloGeom hg) {
  constexpr int NWV = 8;
  constexpr int BN = 32 * NT;
  constexpr int BMH = 32 * MW * NWV;
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int W_INSTR = W_BYTES / 1024;
  constexpr int IPW = (W_INSTR + NWV - 1) / NWV;
  constexpr int MAXR = 7;                 // LDS-DMA rounds a step may carry (the host schedules within it)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [halo buffers x nbuf][W ring x 3]

#define NBDT_PIN(x) __builtin_amdgcn_readfirstlane(x)
  // Persistent blocks: block b sits on XCD b & 7 and walks that XCD's tile range, every class of a tile.
  const int bid = blockIdx.x;
  const int nl = gridDim.x >> 3;
  const int t_lo = (bid & 7) * p.tiles_per_xcd;
  const int t_n = min(p.tiles_per_xcd, p.tiles - t_lo);
  if (t_n <= 0) return;
  const int n_items = t_n * p.ncls;
  int li = bid >> 3;
  if (li >= n_items) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = NBDT_PIN(tid >> 6);
  const int grp = wave >> 2;
  const nbdt_conv_desc& d = p.c.d;

  const int a_bytes = NBDT_PIN(hg.a_bytes);
  const int a_instr = NBDT_PIN(hg.a_instr);
  const int hw2 = NBDT_PIN(hg.hw2), himg = NBDT_PIN(hg.himg);
  const int nbuf = NBDT_PIN(p.nbuf);
  const int last_pix = NBDT_PIN(d.B * (d.gh + 2) * (d.gw + 2) - 1);
  auto pin_ptr = [](const void* q) {
    const unsigned long long u = (unsigned long long)q;
    return (const bf16_t*)(((unsigned long long)NBDT_PIN((unsigned)(u >> 32)) << 32) | (unsigned)NBDT_PIN((unsigned)u));
  };
  // (two input tensors: a step's source is one scalar select away -- an indexed load from the kernel-argument segment
  //  in the MFMA segment put an s_waitcnt into the middle of the MFMA stream, a four-way select became branches)
  const bf16_t* in0 = pin_ptr(p.in[0]);
  const bf16_t* in1 = pin_ptr(p.in[1]);
  const int ps0 = NBDT_PIN(p.pix_stride[0]), ps1 = NBDT_PIN(p.pix_stride[1]);
  const bf16_t* w_all = pin_ptr(p.w_tiles);
  typedef const __attribute__((address_space(4))) int* cint_ptr;       // constant address space: scalar loads
  const cint_ptr steps_all = (cint_ptr)(unsigned long long)p.steps;

  struct Tile {
    int li, cls, m_blk, n_blk, m0, n0, base_pix, nsteps, npro;
    cint_ptr steps;
    const bf16_t* w_tiles;
    int out_bs, out_hs, out_ws, out_base;
    int pro_t0, pro_t1, pro_c0, pro_c1;
  };
  auto tile_of = [&](int it) {
    Tile t;
    t.li = it;
    t.cls = it / t_n;
    const int tile = t_lo + (it - t.cls * t_n);
    t.m_blk = tile / p.c.n_blocks;
    t.n_blk = tile - t.m_blk * p.c.n_blocks;
    t.m0 = t.m_blk * BMH;
    t.n0 = t.n_blk * BN;
    const int img_pp = (d.gh + 2) * (d.gw + 2);
    if (hg.ib == 1) {
      const int b0 = t.m_blk / hg.blocks_per_img;
      t.base_pix = b0 * img_pp + (t.m_blk - b0 * hg.blocks_per_img) * hg.rb * (d.gw + 2);
    } else {
      t.base_pix = t.m_blk * hg.ib * img_pp;
    }
    t.base_pix = NBDT_PIN(t.base_pix);
    const nbdt::SegClassDev& c = p.cls[t.cls];
    t.nsteps = NBDT_PIN(c.nsteps);
    t.npro = NBDT_PIN(c.npro);
    t.steps = steps_all + (size_t)NBDT_PIN(c.step0) * 8;
    t.w_tiles = w_all + c.w_off + (size_t)t.n_blk * t.nsteps * (BN * 32);
    t.out_bs = c.out_bs; t.out_hs = c.out_hs; t.out_ws = c.out_ws; t.out_base = c.out_base;
    t.pro_t0 = c.pro_tensor[0]; t.pro_t1 = c.pro_tensor[1];
    t.pro_c0 = c.pro_choff[0]; t.pro_c1 = c.pro_choff[1];
    return t;
  };
  Tile cur = tile_of(li);
#undef NBDT_PIN

  // ---- DMA addressing (as in conv3x3_pp_kernel): global_load_lds <lane offset>, <SGPR base>; a piece = 16 halo pixels
  // x 64 B, lane -> pixel lane >> 2, LDS chunk position lane & 3 holding source chunk (lane & 3) ^ ((pixel >> 2) & 3).
  int a_lane_pix, a_lane_el;
  unsigned w_voff;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned w_ring = lds_base + nbuf * a_bytes;
  auto issue_w = [&](const bf16_t* w_tiles, int slot, int t) {
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
      const int id = wave + NWV * k;
      if (id < W_INSTR)      // wave-uniform
        glds16_s(w_tiles + (t * W_INSTR + id) * 512, w_voff, w_ring + slot * W_BYTES + id * 1024);
    }
  };

  f32x16 acc[NT][MW];
  int frag_half;
  int hp0[MW];
  unsigned w_rd0, w_rd1;
  auto lane_constants = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    a_lane_pix = ln >> 2;
    a_lane_el = (((ln & 3) ^ ((ln >> 4) & 3)) << 3);
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
      hp0[tm] = img * himg + r * hw2 + c;
    }
    const int w_frag_off = frag_row * 64 + ((frag_half ^ ((frag_row >> 2) & 3)) << 4);
    w_rd0 = nbuf * a_bytes + w_frag_off;
    w_rd1 = nbuf * a_bytes + (w_frag_off ^ 32);
  };
  lane_constants();

  typedef const __attribute__((address_space(3))) unsigned char* lds_cptr;
  const lds_cptr smem3 = (lds_cptr)smem;
  struct Plan {
    lds_cptr ra[2][MW];     // LDS addresses of the pixel fragments [ks][tm]
    unsigned voff0, voffL;  // lane byte offsets of round 0 / of the last round (clamped to the tensor)
    int n;                  // rounds this wave issues
    int strict;
    const bf16_t* src;      // tensor + channel of the slice being fetched
    int round_el;           // elements between rounds: 8 pieces x 16 pixels x pix_stride
    unsigned dst;           // LDS address of this wave's piece of round 0
    int pix, ps;            // ragged tiles: halo pixel of piece id0, elements per pixel of the source tensor
  };
  auto load_step = [&](cint_ptr steps, int t) {
    nbdt::SegStep s;
    const cint_ptr q = steps + t * 8;
    s.toff = q[0]; s.abuf = q[1]; s.pf_n = q[2]; s.pf_dst = q[3];
    s.pf_pix = q[4]; s.pf_choff = q[5]; s.pf_tensor = q[6]; s.strict = q[7];
    return s;
  };
  int base_pix_w = cur.base_pix + wave * 16;     // halo pixel of lane 0 of this wave's piece of round 0, id0 = 0
  auto prepare = [&](const nbdt::SegStep& r) {
    Plan q;
    int h0 = hp0[0], h1 = hp0[MW - 1], lp = a_lane_pix, le = a_lane_el;
    asm volatile("" : "+v"(h0), "+v"(h1), "+v"(lp), "+v"(le));     // the address math stays in the calling segment
    auto frag_off = [&](int h) {
      const int hp = h + r.toff;
      return (unsigned)(hp * 64 + ((frag_half ^ ((hp >> 2) & 3)) << 4));
    };
    {
      const unsigned o = frag_off(h0);
      q.ra[0][0] = smem3 + (r.abuf + o); q.ra[1][0] = smem3 + (r.abuf + (o ^ 32));
    }
    if constexpr (MW == 2) {
      const unsigned o = frag_off(h1);
      q.ra[0][MW - 1] = smem3 + (r.abuf + o); q.ra[1][MW - 1] = smem3 + (r.abuf + (o ^ 32));
    }
    // rounds: pieces id0 + 8 j + wave; only the slice's last round can be short of waves
    // (r.strict >> 8 = waves that issue the step's last round: the slice's last round can be short of waves)
    const int n = r.pf_n - (wave >= (r.strict >> 8) ? 1 : 0);
    q.n = n;
    q.strict = r.strict & 1;
    const int ps = r.pf_tensor ? ps1 : ps0;
    q.src = (r.pf_tensor ? in1 : in0) + r.pf_choff;
    q.round_el = 128 * ps;
    q.dst = lds_base + r.pf_dst + wave * 1024;
    q.pix = r.pf_pix;
    q.ps = ps;
    const int px0 = lp + (base_pix_w + r.pf_pix);
    const int pxa = px0 < last_pix ? px0 : last_pix;
    int pxl = px0 + 128 * (n - 1);
    pxl = pxl < last_pix ? pxl : last_pix;          // tail lanes of the tensor's last piece re-read its last pixel
    q.voff0 = (unsigned)(pxa * ps + le) * 2u;
    q.voffL = (unsigned)(pxl * ps + le) * 2u;
    return q;
  };

  // ---- a tile's first LDS-DMA: its prologue slices (every piece) into buffers 0 .. npro-1, W(0), W(1)
  auto issue_slice = [&](int tensor, int choff, int base_pix, int buf) {
    const int ps = tensor ? ps1 : ps0;
    const bf16_t* src = (tensor ? in1 : in0) + choff;
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int id = wave + NWV * k;
      if (id < a_instr) {
        int px = a_lane_pix + (base_pix + id * 16);
        px = px < last_pix ? px : last_pix;
        glds16_sf(src, (unsigned)(px * ps + a_lane_el) * 2u, lds_base + buf * a_bytes + id * 1024);
      }
    }
  };
  auto issue_first = [&](const Tile& t) {
    issue_slice(t.pro_t0, t.pro_c0, t.base_pix, 0);
    if (t.npro > 1) issue_slice(t.pro_t1, t.pro_c1, t.base_pix, 1);
    issue_w(t.w_tiles, 0, 0);
    if (t.nsteps > 1) issue_w(t.w_tiles, 1, 1);
  };
  // ---- LDS of the epilogue (conv3x3_pp_kernel's scheme).  Packed from smem + 0 it overlaps halo buffer 0 and the ring, so
  // the next tile cannot start before it is done.  With p.overlap (the launch checked that it fits) the per-wave regions
  // go into halo buffers 1 .. nbuf-1 and behind ring slot 1 (or 0: !early_w1), and the next tile's slice 0 and W(0) [W(1)]
  // are issued at the top of the epilogue: their round trip and the block's idle prologue disappear behind it.  A second
  // prologue slice (classes that start with 1-tap slices) and a late W(1) follow after the epilogue.
  constexpr int EPI_REGION = 32 * (2 * BN + 16);
  EpiLds epi_lds = epi_lds_packed<NT, NWV>(smem, wave);
  const bool overlap = p.overlap != 0, early_w1 = p.early_w1 != 0;
  if (overlap) {
    const int n_a = min(NWV, ((nbuf - 1) * a_bytes) / EPI_REGION);
    unsigned char* tail = smem + nbuf * a_bytes + (early_w1 ? 2 : 1) * W_BYTES;
    epi_lds.region = wave < n_a ? smem + a_bytes + wave * EPI_REGION : tail + (wave - n_a) * EPI_REGION;
    epi_lds.row_off = (int*)(tail + (NWV - n_a) * EPI_REGION) + wave * 64;
    epi_lds.blk_stats = (float*)(tail + (NWV - n_a) * EPI_REGION + NWV * 64 * 4);
    epi_lds.base_a = smem + a_bytes; epi_lds.base_b = tail; epi_lds.n_a = n_a;
  }
  auto issue_early = [&](const Tile& t) {
    issue_slice(t.pro_t0, t.pro_c0, t.base_pix, 0);
    issue_w(t.w_tiles, 0, 0);
    if (early_w1 && t.nsteps > 1) issue_w(t.w_tiles, 1, 1);
  };

  issue_first(cur);
  for (;;) {   // ======================================= one (class, output tile) =======================================
    const int m_blk = cur.m_blk, m0 = cur.m0, n0 = cur.n0;
    const bf16_t* w_tiles = cur.w_tiles;
    const cint_ptr steps = cur.steps;
    const int nsteps = cur.nsteps;
    li = cur.li;
    lane_constants();
    base_pix_w = cur.base_pix + wave * 16;
    const bool ragged = cur.base_pix + (a_instr - 1) * 16 > last_pix;      // (more than the last piece reaches past the tensor)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int tm = 0; tm < MW; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;
#if NBDT_SEG_TIMING
    const unsigned tm_top = seg_stamp();
#endif
    Plan plan = prepare(load_step(steps, 0));
    int prev_a = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's first DMA (and the last tile's stores)
    __builtin_amdgcn_s_barrier();            // bP: every wave's pieces have landed; the last epilogue's LDS is free
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#if NBDT_SEG_TIMING
    unsigned tm_l = 0, tm_b1 = 0, tm_m = 0, tm_b2 = 0;
    const unsigned tm_begin = seg_stamp();
    unsigned tm_prev = tm_begin;
#endif

    // one K step with its weight tile in ring slot K (the ring position is the only literal left)
    auto step = [&](auto KC, int t) __attribute__((always_inline)) {
      constexpr int K = decltype(KC)::value;
      // ================= L(t): fragments -> registers, this wave's DMA pieces =================
      const int tn1 = t + 1 < nsteps ? t + 1 : t;
      const nbdt::SegStep nrec = load_step(steps, tn1);      // (s_load: lands under the reads, used in M(t))
      bf16x8 pf[2][MW], wf[2][NT];
#pragma unroll
      for (int tm = 0; tm < MW; ++tm) {
        pf[0][tm] = *(const __attribute__((address_space(3))) bf16x8*)plan.ra[0][tm];
        pf[1][tm] = *(const __attribute__((address_space(3))) bf16x8*)plan.ra[1][tm];
      }
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        wf[0][tn] = *(const bf16x8*)(smem + w_rd0 + (K * W_BYTES + tn * 2048));
        wf[1][tn] = *(const bf16x8*)(smem + w_rd1 + (K * W_BYTES + tn * 2048));
      }
      // the weight pieces this wave issued one step ago must have landed; that step's halo pieces came after them in
      // issue order and may stay in flight unless the schedule marked them strict
      // (at most two stay in flight: a wider switch compiled to a tree of a dozen scalar branches per step)
      // (hand-written: hipcc turns the three-way if into a dozen scalar instructions with mask registers)
      asm volatile(
          "s_cmp_lt_u32 %0, 2\n\t"
          "s_cbranch_scc1 .Lseg_lt2_%=\n\t"
          "s_waitcnt vmcnt(2)\n\t"
          "s_branch .Lseg_done_%=\n"
          ".Lseg_lt2_%=:\n\t"
          "s_cmp_eq_u32 %0, 0\n\t"
          "s_cbranch_scc1 .Lseg_0_%=\n\t"
          "s_waitcnt vmcnt(1)\n\t"
          "s_branch .Lseg_done_%=\n"
          ".Lseg_0_%=:\n\t"
          "s_waitcnt vmcnt(0)\n"
          ".Lseg_done_%=:"
          ::"s"(prev_a) : "memory", "scc");
      if (t + 2 < nsteps) issue_w(w_tiles, (K + 2) % 3, t + 2);
      if (ragged) {
        // a tile that reaches past the end of the tensor (fewer images than a tile holds): every piece clamps its own
        // pixels (the fast path below lets only the slice's last piece overrun)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lp = ln >> 2, le = (((ln & 3) ^ ((ln >> 4) & 3)) << 3);
#pragma unroll
        for (int j = 0; j < MAXR; ++j)
          if (j < plan.n) {
            int px = lp + (base_pix_w + plan.pix + 128 * j);
            px = px < last_pix ? px : last_pix;
            glds16_sf(plan.src, (unsigned)(px * plan.ps + le) * 2u, plan.dst + j * 8192);
          }
      } else if (plan.n > 0) {
        glds16_sf(plan.src, plan.voffL, plan.dst + (plan.n - 1) * 8192);
        const int m = plan.n - 1;       // rounds 0 .. m-1 share one lane offset; nested so that issuing is the fall-through path
#define NBDT_ROUND(J) glds16_sf(plan.src + J * plan.round_el, plan.voff0, plan.dst + J * 8192)
        if (__builtin_expect(m > 0, 1)) { NBDT_ROUND(0);
          if (__builtin_expect(m > 1, 1)) { NBDT_ROUND(1);
            if (m > 2) { NBDT_ROUND(2);
              if (m > 3) { NBDT_ROUND(3);
                if (m > 4) { NBDT_ROUND(4);
                  if (m > 5) { NBDT_ROUND(5); } } } } } }
#undef NBDT_ROUND
      }
      prev_a = plan.strict ? 0 : plan.n;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      NBDT_SEG_STAMP(tm_l)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      NBDT_SEG_STAMP(tm_b1)
      // ================= M(t): 2*MW*NT MFMAs; the idle issue slots between them prepare L(t+1) =================
      __builtin_amdgcn_s_setprio(1);
      plan = prepare(nrec);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int tm = 0; tm < MW; ++tm)
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][tn], pf[ks][tm], acc[tn][tm], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2 * MW * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, MW == 2 ? 2 : 3, 0);   // VALU
        __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);   // SALU
      }
      asm volatile("" : "+v"(plan.ra[0][0]), "+v"(plan.ra[1][0]), "+v"(plan.voff0), "+v"(plan.voffL));
      if (MW == 2) asm volatile("" : "+v"(plan.ra[0][MW - 1]), "+v"(plan.ra[1][MW - 1]));
      __builtin_amdgcn_s_setprio(0);
      NBDT_SEG_STAMP(tm_m)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      NBDT_SEG_STAMP(tm_b2)
    };
    for (int t = 0; t < nsteps; t += 3) {
      step(std::integral_constant<int, 0>{}, t);
      if (t + 1 < nsteps) step(std::integral_constant<int, 1>{}, t + 1);
      if (t + 2 < nsteps) step(std::integral_constant<int, 2>{}, t + 2);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue with the class's own output map
    nbdt::ConvDmaParams pc = p.c;
    pc.d.out_bs = cur.out_bs; pc.d.out_hs = cur.out_hs; pc.d.out_ws = cur.out_ws; pc.d.out_base = cur.out_base;
#if NBDT_SEG_TIMING
    const unsigned tm_epi = seg_stamp();
#endif
    const int next = li + nl;
    const bool more = next < n_items;
    const Tile nxt = more ? tile_of(next) : cur;
    const bool early = more && overlap;
    auto hook = [&]() {
      if (early) issue_early(nxt);
    };
    conv_epilogue<NT, HAS_RES, STATS, NWV, MW>(acc, pc, epi_lds, m0, n0, m_blk, wave, lane, tid, nullptr, hook);
#if NBDT_SEG_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const int gi = ((bid & 7) * p.tiles_per_xcd * p.ncls + li) & 1023;      // (XCD-major item index, first 1024)
      unsigned* o = g_seg_timing + (gi * 8 + wave) * 8;
      o[0] = tm_l; o[1] = tm_b1; o[2] = tm_m; o[3] = tm_b2; o[4] = tm_epi - tm_begin; o[5] = nsteps;
      o[6] = tm_begin - tm_top; o[7] = seg_stamp() - tm_epi;
    }
#endif

    if (!more) break;
    cur = nxt;
    if (early) {
      const bool late_slice = cur.npro > 1, late_w1 = !early_w1 && cur.nsteps > 1;
      if (late_slice || late_w1) {      // they land in LDS the epilogue used: every wave must be out of it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (late_slice) issue_slice(cur.pro_t1, cur.pro_c1, cur.base_pix, 1);
        if (late_w1) issue_w(cur.w_tiles, 1, 1);
      }
      continue;
    }
    // the packed epilogue LDS overlaps halo buffer 0 and the ring: every wave must be out of it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_first(cur);
  }
}

// ------------------------------------------------------------------------------------------------------------
// weights -> DMA-ordered tiles: one block per (class, cout tile, K step); a tile is the swizzled LDS image of
// (32*NT rows) x 32 k (conv_common.h lds_off: 16-byte chunk position cp of row r holds chunk cp ^ ((r >> 2) & 3)).
struct SegTileArgs {
  const bf16_t* w[4];
  int row_stride[4];
  long long cls_w_off[4];
  int cls_step0[4], cls_nsteps[4];
  int ncls, n_blocks, bn;
};
__global__ __launch_bounds__(256) void conv_seg_tile_kernel(SegTileArgs a, const nbdt::SegWStep* __restrict__ wsteps,
                                                            bf16_t* __restrict__ dst) {
  int e = blockIdx.x, c = 0;
  while (c + 1 < a.ncls && e >= a.n_blocks * a.cls_nsteps[c]) { e -= a.n_blocks * a.cls_nsteps[c]; ++c; }
  const int n_blk = e / a.cls_nsteps[c], t = e - n_blk * a.cls_nsteps[c];
  const nbdt::SegWStep ws = wsteps[a.cls_step0[c] + t];
  const bf16_t* src = seg_sel4(ws.matrix, a.w[0], a.w[1], a.w[2], a.w[3]);
  const long long rs = seg_sel4(ws.matrix, a.row_stride[0], a.row_stride[1], a.row_stride[2], a.row_stride[3]);
  bf16_t* d0 = dst + a.cls_w_off[c] + ((long long)n_blk * a.cls_nsteps[c] + t) * (a.bn * 32);
  for (int q = threadIdx.x; q < a.bn * 4; q += 256) {
    const int r = q >> 2, cp = q & 3;
    const int ch = cp ^ ((r >> 2) & 3);
    const u32x4_t v = *(const u32x4_t*)(src + (long long)(n_blk * a.bn + r) * rs + ws.w_off + ch * 8);
    *(u32x4_t*)(d0 + r * 32 + cp * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------------------
// VERIFICATION-ONLY fp32 twin (one thread per output element; see ref_fp32.hip for why these exist)
struct SegRefArgs {
  const float* in[4];
  const float* w[4];
  int pix_stride[4], row_stride[4];
  int B, gh, gw, cout;
  int step0, nsteps;
  int out_bs, out_hs, out_ws, out_base;
};
__global__ __launch_bounds__(256) void conv_seg_ref_kernel(SegRefArgs a, const nbdt::SegRefStep* __restrict__ rs,
                                                           float* __restrict__ out, const float* __restrict__ res,
                                                           long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx % a.cout), m = (int)(idx / a.cout);
  const int x = m % a.gw, t0 = m / a.gw, y = t0 % a.gh, b = t0 / a.gh;
  const int po = b * a.out_bs + y * a.out_hs + x * a.out_ws + a.out_base + n;
  float acc = 0.f;
  for (int t = 0; t < a.nsteps; ++t) {
    const nbdt::SegRefStep s = rs[a.step0 + t];
    const int ps = seg_sel4(s.tensor, a.pix_stride[0], a.pix_stride[1], a.pix_stride[2], a.pix_stride[3]);
    const float* in = seg_sel4(s.tensor, a.in[0], a.in[1], a.in[2], a.in[3]);
    const float* w = seg_sel4(s.matrix, a.w[0], a.w[1], a.w[2], a.w[3]);
    const long long wrs = seg_sel4(s.matrix, a.row_stride[0], a.row_stride[1], a.row_stride[2], a.row_stride[3]);
    const float* pa = in + ((long long)(b * (a.gh + 2) + y + s.dy) * (a.gw + 2) + x + s.dx) * ps + s.ch0;
    const float* pw = w + (long long)n * wrs + s.w_off;
    float sum = 0.f;
    for (int k = 0; k < 32; ++k) sum += pa[k] * pw[k];
    acc += sum;
  }
  if (res) acc += res[po];
  out[po] = acc;
}

namespace nbdt {

// ------------------------------------------------------------------------------------------------------------
// host side: the plan
struct SegPlan {
  nbdt_conv_seg_desc d;
  std::vector<nbdt_conv_seg_slice> slices[4];
  HaloGeom hg;
  int tile, nbuf, nt, max_rounds;
  long long w_elems;
  std::vector<SegStep> steps;
  std::vector<SegWStep> wsteps;
  std::vector<SegRefStep> rsteps;
  SegClassDev cls[4];
  struct Dev { SegStep* steps; SegWStep* wsteps; SegRefStep* rsteps; };
  std::mutex m;
  std::map<int, Dev> dev;
};

static int seg_cout_tile(int cout) {
  const int nt32 = cout / 32;
  return nt32 % 5 == 0 ? 5 : (nt32 % 4 == 0 ? 4 : (nt32 % 2 == 0 ? 2 : 1));
}

// halo geometry of `tile` pixels (whole image rows or whole images), contiguous LDS image (no pitch padding)
static bool seg_geom(int gh, int gw, int tile, HaloGeom* hg) {
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
  hg->pad = 0; hg->dv = 0;
  hg->lpitch = hg->hw2;
  hg->limg = hg->himg;
  hg->row_magic = (65536 + hg->lpitch - 1) / hg->lpitch;
  hg->a_instr = (hg->hp * 4 + 63) / 64;
  hg->a_bytes = hg->a_instr * 1024;
  hg->blocks_per_img = ib == 1 ? gh / rb : 1;
  hg->nwv = 8;
  hg->mw = tile == 256 ? 1 : 2;
  return true;
}

// Spread the LDS-DMA rounds of every slice of one class over the K steps.  ntaps[s] = steps of slice s; R = rounds per
// slice (8 pieces each); slice s may be issued in steps [lo, hi]: lo = the step after the last read of the buffer's
// previous tenant (slice s - nbuf), hi = two steps before its first read.  Slices with an empty window at the start of
// the class go into the tile's prologue.  One slice per step; minimise the sum of squared rounds per step (dynamic
// programme over (slice, its last issue step)); a step that issues at hi is strict.  Returns false if impossible.
struct SegIssue { int slice, round0, n, strict; };
static bool seg_schedule(const std::vector<int>& ntaps, int nbuf, int R, int maxr, std::vector<SegIssue>* issue,
                         int* npro_out) {
  const int S = (int)ntaps.size();
  std::vector<int> first(S), last(S);
  int T = 0;
  for (int s = 0; s < S; ++s) { first[s] = T; T += ntaps[s]; last[s] = T - 1; }
  issue->assign(T, SegIssue{-1, 0, 0, 0});
  std::vector<int> lo(S), hi(S);
  int npro = 1;
  for (int s = 1; s < S; ++s) {
    lo[s] = s >= nbuf ? last[s - nbuf] + 1 : 0;
    hi[s] = first[s] - 2;
    if (hi[s] < lo[s]) {
      if (s == npro && s < nbuf && s < 2) { ++npro; continue; }
      return false;
    }
  }
  *npro_out = npro;
  if (npro >= S) return true;
  const double INF = 1e30;
  auto cost = [&](int k) -> double {      // R rounds spread evenly over k steps
    if ((R + k - 1) / k > maxr) return INF;
    const int q = R / k, rem = R % k;
    return (double)rem * (q + 1) * (q + 1) + (double)(k - rem) * q * q;
  };
  // dp[j][e]: slices npro..j issued, slice j's last issue step is e
  std::vector<std::vector<double>> dp(S, std::vector<double>(T, INF));
  std::vector<std::vector<int>> from_s(S, std::vector<int>(T, -1)), from_e(S, std::vector<int>(T, -1));
  for (int j = npro; j < S; ++j) {
    // prefix minimum of the previous slice's table: best[x] = min over e' <= x
    std::vector<double> best(T + 1, INF);
    std::vector<int> best_e(T + 1, -1);
    if (j > npro) {
      for (int x = 0; x < T; ++x) {
        best[x + 1] = best[x]; best_e[x + 1] = best_e[x];
        if (dp[j - 1][x] < best[x + 1]) { best[x + 1] = dp[j - 1][x]; best_e[x + 1] = x; }
      }
    }
    for (int e = lo[j]; e <= hi[j]; ++e)
      for (int s = lo[j]; s <= e; ++s) {
        double prev = 0.0;
        int pe = -1;
        if (j > npro) { prev = best[s]; pe = best_e[s]; if (prev >= INF) continue; }      // previous slice ends before s
        const double c = cost(e - s + 1);
        if (c >= INF) continue;
        const double v = prev + c + (e == hi[j] ? 0.5 : 0.0);
        if (v < dp[j][e]) { dp[j][e] = v; from_s[j][e] = s; from_e[j][e] = pe; }
      }
  }
  int e = -1;
  double bv = INF;
  for (int x = 0; x < T; ++x) if (dp[S - 1][x] < bv) { bv = dp[S - 1][x]; e = x; }
  if (e < 0) return false;
  for (int j = S - 1; j >= npro; --j) {
    const int s = from_s[j][e], pe = from_e[j][e];
    const int k = e - s + 1, q = R / k, rem = R % k;
    int r0 = 0;
    for (int i = 0; i < k; ++i) {
      const int n = q + (i < rem ? 1 : 0);
      if (n > 0) (*issue)[s + i] = SegIssue{j, r0, n, (s + i) == hi[j] ? 1 : 0};
      r0 += n;
    }
    e = pe;
  }
  return true;
}

static int seg_build(SegPlan* P, int tile, int nbuf) {
  const nbdt_conv_seg_desc& d = P->d;
  HaloGeom hg;
  if (!seg_geom(d.gh, d.gw, tile, &hg)) return 1;
  const int nt = seg_cout_tile(d.cout);
  const int R = (hg.a_instr + 7) / 8;
  if (R > 7) return 1;
  const size_t lds = (size_t)nbuf * hg.a_bytes + (size_t)3 * nt * 32 * BK * 2;
  if (lds > 160 * 1024) return 1;
  P->steps.clear(); P->wsteps.clear(); P->rsteps.clear();
  P->max_rounds = 0;
  long long w_off = 0;
  const int n_blocks = d.cout / (32 * nt);
  for (int c = 0; c < d.nclasses; ++c) {
    const std::vector<nbdt_conv_seg_slice>& sl = P->slices[c];
    std::vector<int> ntaps;
    for (const auto& s : sl) ntaps.push_back(s.ntaps);
    std::vector<SegIssue> issue;
    int npro = 1;
    if (!seg_schedule(ntaps, nbuf, R, 7, &issue, &npro)) return 1;
    SegClassDev& cd = P->cls[c];
    memset(&cd, 0, sizeof(cd));
    cd.step0 = (int)P->steps.size();
    cd.npro = npro;
    cd.w_off = w_off;
    cd.out_bs = d.cls[c].out_bs; cd.out_hs = d.cls[c].out_hs; cd.out_ws = d.cls[c].out_ws; cd.out_base = d.cls[c].out_base;
    for (int i = 0; i < npro && i < 2; ++i) { cd.pro_tensor[i] = sl[i].tensor; cd.pro_choff[i] = sl[i].ch0; }
    int t = 0;
    for (int s = 0; s < (int)sl.size(); ++s)
      for (int j = 0; j < sl[s].ntaps; ++j, ++t) {
        SegStep r;
        const int tap = sl[s].tap[j];
        r.toff = (tap / 3) * hg.hw2 + (tap % 3);
        r.abuf = (s % nbuf) * hg.a_bytes;
        const SegIssue& is = issue[t];
        r.pf_n = is.slice >= 0 ? is.n : 0;
        r.strict = is.slice >= 0 ? is.strict : 0;
        {   // waves that issue the step's last round (pieces id0 + 8 (n - 1) + wave < a_instr), in bits 8..
          const int last_waves = r.pf_n > 0 ? std::min(8, hg.a_instr - (is.round0 + r.pf_n - 1) * 8) : 8;
          r.strict |= last_waves << 8;
        }
        const int fs = is.slice >= 0 ? is.slice : 0;
        r.pf_dst = (fs % nbuf) * hg.a_bytes + is.round0 * 8 * 1024;
        r.pf_pix = is.round0 * 8 * 16;
        r.pf_choff = sl[fs].ch0;
        r.pf_tensor = sl[fs].tensor;
        P->steps.push_back(r);
        P->wsteps.push_back(SegWStep{sl[s].w_matrix, sl[s].w_off[j]});
        P->rsteps.push_back(SegRefStep{sl[s].tensor, sl[s].ch0, tap / 3, tap % 3, sl[s].w_matrix, sl[s].w_off[j]});
        P->max_rounds = std::max(P->max_rounds, r.pf_n);
      }
    cd.nsteps = t;
    w_off += (long long)n_blocks * t * (32 * nt) * 32;
  }
  P->w_elems = w_off;
  P->hg = hg; P->tile = tile; P->nbuf = nbuf; P->nt = nt;
  return 0;
}

static int seg_dev(SegPlan* P, SegPlan::Dev* out) {
  int dev = 0;
  NBDT_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(P->m);
  auto it = P->dev.find(dev);
  if (it != P->dev.end()) { *out = it->second; return NBDT_OK; }
  SegPlan::Dev dv{nullptr, nullptr, nullptr};
  NBDT_HIP_CHECK(hipMalloc((void**)&dv.steps, P->steps.size() * sizeof(SegStep)));
  NBDT_HIP_CHECK(hipMalloc((void**)&dv.wsteps, P->wsteps.size() * sizeof(SegWStep)));
  NBDT_HIP_CHECK(hipMalloc((void**)&dv.rsteps, P->rsteps.size() * sizeof(SegRefStep)));
  NBDT_HIP_CHECK(hipMemcpy(dv.steps, P->steps.data(), P->steps.size() * sizeof(SegStep), hipMemcpyHostToDevice));
  NBDT_HIP_CHECK(hipMemcpy(dv.wsteps, P->wsteps.data(), P->wsteps.size() * sizeof(SegWStep), hipMemcpyHostToDevice));
  NBDT_HIP_CHECK(hipMemcpy(dv.rsteps, P->rsteps.data(), P->rsteps.size() * sizeof(SegRefStep), hipMemcpyHostToDevice));
  P->dev[dev] = dv;
  *out = dv;
  return NBDT_OK;
}

template <int NT, int MW>
static int seg_launch(SegPlan* P, ConvSegParams& p, hipStream_t st) {
  constexpr int NWV = 8;
  constexpr size_t W_BYTES = (size_t)NT * 32 * BK * 2;
  constexpr size_t REGION = 32 * (2 * 32 * NT + 16), TABLES = NWV * 64 * 4 + 2 * 32 * NT * 4;
  const size_t bufs = (size_t)P->nbuf * P->hg.a_bytes;
  size_t shmem = bufs + 3 * W_BYTES;
  const size_t epi = conv_epilogue_lds_bytes<NT, NWV>();
  if (shmem < epi) shmem = epi;
  // the epilogue's regions in halo buffers 1 .. nbuf-1 and behind ring slot 1 (W(0), W(1) early) or slot 0 (W(1) late)
  p.overlap = 0; p.early_w1 = 0;
  {
    const size_t n_a = std::min<size_t>(NWV, ((size_t)(P->nbuf - 1) * P->hg.a_bytes) / REGION);
    for (int early = 1; early >= 0 && !p.overlap; --early) {
      const size_t need = bufs + (early ? 2 : 1) * W_BYTES + (NWV - n_a) * REGION + TABLES;
      if (need <= 160 * 1024) {
        p.overlap = 1; p.early_w1 = early;
        if (shmem < need) shmem = need;
      }
    }
    if (getenv("NBDT_SEG_NO_OVERLAP")) p.overlap = 0;       // (A/B)
  }
  const int per_xcd_items = p.tiles_per_xcd * p.ncls;
  const int per_round = std::min(per_xcd_items, std::max(1, 32 - (reserved_cus() + 7) / 8));
  static DeviceAttr site;
#define NBDT_KERNEL(R, S) reinterpret_cast<const void*>(&conv_seg_kernel<NT, R, S, MW>)
  if (site.need(shmem)) {
#define NBDT_ATTR(R, S) \
  NBDT_ATTR_CHECK(site, hipFuncSetAttribute(NBDT_KERNEL(R, S), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem))
    NBDT_ATTR(false, 0); NBDT_ATTR(false, 1); NBDT_ATTR(true, 0); NBDT_ATTR(true, 1);
#undef NBDT_ATTR
    site.done(shmem);
  }
  const dim3 grid(per_round * 8), blk(64 * NWV);
  void* args[] = {(void*)&p, (void*)&P->hg};
#define NBDT_GO(R, S)                                                                                                \
  do {                                                                                                                  \
    snprintf(g_last_igemm_full, sizeof(g_last_igemm_full), "conv_seg_kernel<%d, %s, %d, %d>", NT, R ? "true" : "false", S, MW); \
    NBDT_HIP_CHECK(hipLaunchKernel(NBDT_KERNEL(R, S), grid, blk, args, shmem, st));                                  \
  } while (0)
  if (p.c.res != nullptr) { if (p.c.stats) NBDT_GO(true, 1); else NBDT_GO(true, 0); }
  else { if (p.c.stats) NBDT_GO(false, 1); else NBDT_GO(false, 0); }
#undef NBDT_GO
#undef NBDT_KERNEL
  return NBDT_OK;
}

}  // namespace nbdt

using nbdt::SegPlan;

extern "C" int nbdt_conv_seg_create(const nbdt_conv_seg_desc* d, void** plan) {
  NBDT_REQUIRE(d && plan, "null argument");
  NBDT_REQUIRE(d->B > 0 && d->gh > 0 && d->gw > 0, "empty pixel grid");
  NBDT_REQUIRE(d->cout > 0 && d->cout % 32 == 0, "cout must be a multiple of 32");
  NBDT_REQUIRE(d->ntensors >= 1 && d->ntensors <= 4 && d->nmatrices >= 1 && d->nmatrices <= 4, "1..4 tensors / matrices");
  NBDT_REQUIRE(d->nclasses >= 1 && d->nclasses <= 4, "1..4 classes");
  NBDT_REQUIRE(d->tile == 0 || d->tile == 256 || d->tile == 512, "tile: 0, 256 or 512");
  NBDT_REQUIRE(d->nbuf == 0 || d->nbuf == 2 || d->nbuf == 3, "nbuf: 0, 2 or 3");
  const long long pix = (long long)d->B * (d->gh + 2) * (d->gw + 2);
  NBDT_REQUIRE((long long)d->B * d->gh * d->gw < (1ll << 31), "pixel grid too large");
  for (int i = 0; i < d->ntensors; ++i) {
    NBDT_REQUIRE(d->pix_stride[i] >= 32 && d->pix_stride[i] % 8 == 0, "pix_stride: >= 32, multiple of 8");
    NBDT_REQUIRE(pix * d->pix_stride[i] * 2 < (1ll << 32), "input tensor beyond 32-bit byte offsets");
  }
  for (int i = 0; i < d->nmatrices; ++i) NBDT_REQUIRE(d->w_row_stride[i] >= 32 && d->w_row_stride[i] % 8 == 0, "w_row_stride");
  bool all_multi = true;        // every slice but a class's last has >= 2 taps: two halo buffers are enough
  for (int c = 0; c < d->nclasses; ++c) {
    const nbdt_conv_seg_class& k = d->cls[c];
    NBDT_REQUIRE(k.nslices >= 1 && k.nslices <= NBDT_SEG_MAX_SLICES && k.slices, "1..NBDT_SEG_MAX_SLICES slices per class");
    NBDT_REQUIRE((k.out_base % 4) == 0 && (k.out_ws % 4) == 0 && (k.out_hs % 4) == 0 && (k.out_bs % 4) == 0,
                 "output pixel offsets must be 8-byte aligned");
    for (int s = 0; s < k.nslices; ++s) {
      const nbdt_conv_seg_slice& sl = k.slices[s];
      NBDT_REQUIRE(sl.tensor >= 0 && sl.tensor < d->ntensors && sl.w_matrix >= 0 && sl.w_matrix < d->nmatrices, "bad tensor / matrix index");
      NBDT_REQUIRE(sl.ch0 >= 0 && sl.ch0 % 8 == 0 && sl.ch0 + 32 <= d->pix_stride[sl.tensor], "bad slice channel");
      NBDT_REQUIRE(sl.ntaps >= 1 && sl.ntaps <= 9, "1..9 taps per slice");
      for (int t = 0; t < sl.ntaps; ++t) {
        NBDT_REQUIRE(sl.tap[t] >= 0 && sl.tap[t] < 9, "tap: 3*R + S in 0..8");
        NBDT_REQUIRE(sl.w_off[t] >= 0 && sl.w_off[t] % 8 == 0 && sl.w_off[t] + 32 <= d->w_row_stride[sl.w_matrix], "bad weight offset");
      }
      if (sl.ntaps < 2 && s + 1 < k.nslices) all_multi = false;
    }
  }
  SegPlan* P = new SegPlan();
  P->d = *d;
  for (int c = 0; c < d->nclasses; ++c) {
    P->slices[c].assign(d->cls[c].slices, d->cls[c].slices + d->cls[c].nslices);
    P->d.cls[c].slices = P->slices[c].data();
  }
  // tile: 512 pixels when that gives at least 3/4 of the CUs an item (every class of a tile is one), else 256
  const int nt = nbdt::seg_cout_tile(d->cout);
  const long long M = (long long)d->B * d->gh * d->gw;
  const long long items512 = (M + 511) / 512 * (d->cout / (32 * nt));
  std::vector<int> tiles, bufs;
  if (d->tile) tiles.push_back(d->tile);
  else if (items512 >= 192) { tiles.push_back(512); tiles.push_back(256); }
  else { tiles.push_back(256); tiles.push_back(512); }
  if (d->nbuf) bufs.push_back(d->nbuf);
  else if (all_multi) { bufs.push_back(2); bufs.push_back(3); }
  else bufs.push_back(3);
  for (int tl : tiles)
    for (int nb : bufs)
      if (nbdt::seg_build(P, tl, nb) == 0) { *plan = P; return NBDT_OK; }
  delete P;
  return nbdt::fail(NBDT_EINVAL, "%s%s", "nbdt_conv_seg_create: no tile / buffer choice fits (tiles must be whole rows or whole images, ",
                    "halo buffers + weight ring within 160 KB, every slice's pieces schedulable)");
}

extern "C" int nbdt_conv_seg_destroy(void* plan) {
  if (!plan) return NBDT_OK;
  SegPlan* P = (SegPlan*)plan;
  for (auto& kv : P->dev) {
    (void)hipFree(kv.second.steps); (void)hipFree(kv.second.wsteps); (void)hipFree(kv.second.rsteps);
  }
  delete P;
  return NBDT_OK;
}

extern "C" int nbdt_conv_seg_info(void* plan, int32_t* tile, int32_t* nbuf, int32_t* steps, int32_t* max_rounds,
                                  int64_t* w_tile_elems) {
  NBDT_REQUIRE(plan, "null plan");
  SegPlan* P = (SegPlan*)plan;
  if (tile) *tile = P->tile;
  if (nbuf) *nbuf = P->nbuf;
  if (steps) for (int c = 0; c < 4; ++c) steps[c] = c < P->d.nclasses ? P->cls[c].nsteps : 0;
  if (max_rounds) *max_rounds = P->max_rounds;
  if (w_tile_elems) *w_tile_elems = P->w_elems;
  return NBDT_OK;
}

// host copy of a class's step records (tests: the schedule is checked on the CPU); returns the number of steps
extern "C" int nbdt_conv_seg_steps(void* plan, int32_t cls, int32_t* out8, int32_t max_steps, int32_t* npro) {
  NBDT_REQUIRE(plan && out8, "null argument");
  SegPlan* P = (SegPlan*)plan;
  NBDT_REQUIRE(cls >= 0 && cls < P->d.nclasses, "bad class");
  const nbdt::SegClassDev& c = P->cls[cls];
  NBDT_REQUIRE(c.nsteps <= max_steps, "buffer too small");
  memcpy(out8, P->steps.data() + c.step0, (size_t)c.nsteps * sizeof(nbdt::SegStep));
  if (npro) *npro = c.npro;
  return c.nsteps;
}

extern "C" int nbdt_conv_seg_tile_weights(void* plan, const void* const* w, void* w_tiles, void* stream) {
  NBDT_REQUIRE(plan && w && w_tiles, "null argument");
  SegPlan* P = (SegPlan*)plan;
  SegPlan::Dev dv;
  int rc = nbdt::seg_dev(P, &dv);
  if (rc) return rc;
  SegTileArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < P->d.nmatrices; ++i) {
    NBDT_REQUIRE(w[i], "null weight matrix");
    a.w[i] = (const bf16_t*)w[i];
    a.row_stride[i] = P->d.w_row_stride[i];
  }
  a.ncls = P->d.nclasses;
  a.bn = 32 * P->nt;
  a.n_blocks = P->d.cout / a.bn;
  long long blocks = 0;
  for (int c = 0; c < a.ncls; ++c) {
    a.cls_w_off[c] = P->cls[c].w_off; a.cls_step0[c] = P->cls[c].step0; a.cls_nsteps[c] = P->cls[c].nsteps;
    blocks += (long long)a.n_blocks * P->cls[c].nsteps;
  }
  hipLaunchKernelGGL(conv_seg_tile_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, dv.wsteps,
                     (bf16_t*)w_tiles);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_conv_seg(void* plan, const void* const* in, const void* w_tiles, void* out, const void* residual,
                             float* bn_partials, void* stream) {
  NBDT_REQUIRE(plan && in && w_tiles && out, "null argument");
  SegPlan* P = (SegPlan*)plan;
  const nbdt_conv_seg_desc& d = P->d;
  NBDT_REQUIRE(!(bn_partials && d.nclasses != 1), "fused statistics are for single-class launches");
  SegPlan::Dev dv;
  int rc = nbdt::seg_dev(P, &dv);
  if (rc) return rc;
  nbdt::ConvSegParams p;
  memset(&p, 0, sizeof(p));
  p.c.d.B = d.B; p.c.d.gh = d.gh; p.c.d.gw = d.gw; p.c.d.cout = d.cout; p.c.d.cin = 32;
  p.c.out = (bf16_t*)out;
  p.c.res = (const bf16_t*)residual;
  p.c.stats = bn_partials;
  p.c.M = d.B * d.gh * d.gw;
  p.c.n_blocks = d.cout / (32 * P->nt);
  p.c.m_blocks = (p.c.M + P->tile - 1) / P->tile;
  p.c.deterministic = nbdt::deterministic() ? 1 : 0;
  for (int i = 0; i < 4; ++i) {
    const int j = i < d.ntensors ? i : 0;
    NBDT_REQUIRE(in[j], "null input tensor");
    p.in[i] = (const bf16_t*)in[j];
    p.pix_stride[i] = d.pix_stride[j];
  }
  p.steps = dv.steps;
  p.w_tiles = (const bf16_t*)w_tiles;
  for (int c = 0; c < d.nclasses; ++c) p.cls[c] = P->cls[c];
  p.ncls = d.nclasses;
  p.nbuf = P->nbuf;
  p.tiles = p.c.m_blocks * p.c.n_blocks;
  p.tiles_per_xcd = (p.tiles + 7) / 8;
  hipStream_t st = (hipStream_t)stream;
  nbdt::g_last_igemm = P->tile == 512 ? "conv_seg_kernel" : "conv_seg_kernel/half";
#define NBDT_DISPATCH(MW)                                         \
  {                                                               \
    if (P->nt == 5) return nbdt::seg_launch<5, MW>(P, p, st);     \
    if (P->nt == 4) return nbdt::seg_launch<4, MW>(P, p, st);     \
    if (P->nt == 2) return nbdt::seg_launch<2, MW>(P, p, st);     \
    return nbdt::seg_launch<1, MW>(P, p, st);                     \
  }
  if (P->tile == 512) NBDT_DISPATCH(2)
  NBDT_DISPATCH(1)
#undef NBDT_DISPATCH
}

extern "C" int nbdt_ref_conv_seg(void* plan, const float* const* in, const float* const* w, float* out,
                                 const float* residual, void* stream) {
  NBDT_REQUIRE(plan && in && w && out, "null argument");
  SegPlan* P = (SegPlan*)plan;
  const nbdt_conv_seg_desc& d = P->d;
  SegPlan::Dev dv;
  int rc = nbdt::seg_dev(P, &dv);
  if (rc) return rc;
  for (int c = 0; c < d.nclasses; ++c) {
    SegRefArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < d.ntensors; ++i) { a.in[i] = in[i]; a.pix_stride[i] = d.pix_stride[i]; }
    for (int i = 0; i < d.nmatrices; ++i) { a.w[i] = w[i]; a.row_stride[i] = d.w_row_stride[i]; }
    a.B = d.B; a.gh = d.gh; a.gw = d.gw; a.cout = d.cout;
    a.step0 = P->cls[c].step0; a.nsteps = P->cls[c].nsteps;
    a.out_bs = d.cls[c].out_bs; a.out_hs = d.cls[c].out_hs; a.out_ws = d.cls[c].out_ws; a.out_base = d.cls[c].out_base;
    const long long total = (long long)d.B * d.gh * d.gw * d.cout;
    hipLaunchKernelGGL(conv_seg_ref_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                       dv.rsteps, out, residual, total);
    NBDT_LAUNCH_CHECK();
  }
  return NBDT_OK;
}
