// Train-mode BatchNorm2d (+ReLU, +residual, +global-avg-pool head) forward/backward on gfx950.
//
// Replaces nn.BatchNorm2d / F.relu / `out += shortcut(x)` / F.avg_pool2d and their autograd on the
// backbone path (reference nbdt/models/resnet.py:69-74, 136-144; pytorchcv PreResUnit/PreResActivation
// behind nbdt/models/wideresnet.py:1-5).  All kernels are HBM-bound streaming passes over padded
// NHWC bf16 tensors: 16-byte (8-channel) vectors per lane, a thread keeps a FIXED channel chunk and
// walks pixels, so per-channel parameters live in registers and per-channel reductions need no
// cross-lane traffic until one LDS fold per block.  Per-channel sums go through NBDT_BN_SLOTS
// replicated accumulators (atomics spread over 32 slots -> no same-address pile-up) and are folded
// by a 1-block finalize kernel; fp32 throughout.
//
// Roofline: HBM.  Algorithmic bytes per element: stats 2 (read x); apply 4 (+2 residual);
// bwd_reduce 6; bwd_apply 8 (+2 gx_add, +2 g_resid).
#include "common.h"

using namespace nbdt;

constexpr int kSlots = NBDT_BN_SLOTS;
constexpr int kMaxThreads = 256;

struct Layout {  // thread layout for a C-channel tensor: C8 channel chunks x PY pixel rows
  int c8, py, threads;
};
static Layout layout_for(int C) {
  Layout l;
  l.c8 = C / 8;
  l.py = kMaxThreads / l.c8;
  if (l.py < 1) l.py = 1;
  l.threads = l.c8 * l.py;
  return l;
}
static int grid_for(const PadGeom& g, const Layout& l, int pixels_per_thread) {
  long long want = ((long long)g.npix + (long long)l.py * pixels_per_thread - 1) / ((long long)l.py * pixels_per_thread);
  if (want < 1) want = 1;
  if (want > 2048) want = 2048;
  return (int)want;
}

// fold per-thread 8-channel partial sums across the PY pixel rows of the block, then add them
// to slot (blockIdx & slot_mask) of scratch[slot][which][C].  slot_mask = kSlots - 1: the 32 replicated accumulators
// (several blocks add to one slot, in whatever order they finish); slot_mask = ~0 (deterministic mode): `scratch` is
// a zeroed row per BLOCK, every address receives exactly one add, and det_fold() sums the rows in block order.
template <int NQ>
__device__ __forceinline__ void block_fold_to_slots(float (&acc)[NQ][8], int cx, int py, int c8, int PY, int C,
                                                    float* scratch, float* lds, unsigned slot_mask) {
  // lds: [PY][c8][NQ*8]
  float* mine = lds + ((size_t)py * c8 + cx) * (NQ * 8);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int i = 0; i < 8; ++i) mine[q * 8 + i] = acc[q][i];
  __syncthreads();
  // output o = q * (c8*8) + channel goes to thread o % (c8*PY): consecutive threads add to consecutive addresses (one
  // 256-byte run per wave; thread (cx, py) folding element e = py, py+PY, ... of ITS chunk put a wave's lanes 32 bytes
  // apart -- 64 sectors per atomic instruction once c8 >= 64).  Each sum is its PY partials in ascending row order.
  const int nch = c8 * 8, nthr = c8 * PY;
  for (int o = py * c8 + cx; o < NQ * nch; o += nthr) {
    const int q = o / nch, c = o - q * nch;
    const float* src = lds + (size_t)(c >> 3) * (NQ * 8) + q * 8 + (c & 7);
    float s = 0.f;
    for (int r = 0; r < PY; ++r) s += src[(size_t)r * c8 * (NQ * 8)];
    atomicAdd(scratch + ((size_t)(blockIdx.x & slot_mask) * NQ + q) * C + c, s);
  }
}

__global__ __launch_bounds__(kMaxThreads) void bn_stats_kernel(const bf16_t* __restrict__ x, PadGeom g, int c8,
                                                               int PY, float* __restrict__ scratch,
                                                               unsigned slot_mask) {
  extern __shared__ float lds[];
  const int cx = threadIdx.x % c8, py = threadIdx.x / c8;
  float acc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.f;
  for (int p = blockIdx.x * PY + py; p < g.npix; p += gridDim.x * PY) {
    const u32x4_t v = *(const u32x4_t*)(x + pad_offset(g, p) + cx * 8);
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[0][i] += f[i];
      acc[1][i] += f[i] * f[i];
    }
  }
  block_fold_to_slots<2>(acc, cx, py, c8, PY, g.C, scratch, lds, slot_mask);
}

// 1 block: fold slots; mode 0 = forward statistics, mode 1 = backward sums
__global__ __launch_bounds__(256) void bn_finalize_kernel(float* __restrict__ scratch, int C, float n,
                                                          float eps, float momentum,
                                                          float* __restrict__ running_mean,
                                                          float* __restrict__ running_var,
                                                          float* __restrict__ save_mean,
                                                          float* __restrict__ save_rstd) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    float s = 0.f, sq = 0.f;
#pragma unroll 8
    for (int k = 0; k < kSlots; ++k) {
      s += scratch[((size_t)k * 2 + 0) * C + c];
      sq += scratch[((size_t)k * 2 + 1) * C + c];
    }
#pragma unroll 8
    for (int k = 0; k < kSlots; ++k) {
      scratch[((size_t)k * 2 + 0) * C + c] = 0.f;   // leave the slots zeroed for the next user
      scratch[((size_t)k * 2 + 1) * C + c] = 0.f;
    }
    const float mean = s / n;
    float var = sq / n - mean * mean;
    var = var > 0.f ? var : 0.f;
    save_mean[c] = mean;
    save_rstd[c] = rsqrtf(var + eps);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      const float unbiased = n > 1.f ? var * n / (n - 1.f) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(float* __restrict__ scratch, int C,
                                                              float* __restrict__ dsum,
                                                              float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
    for (int k = 0; k < kSlots; ++k) {
      s0 += scratch[((size_t)k * 2 + 0) * C + c];
      s1 += scratch[((size_t)k * 2 + 1) * C + c];
    }
#pragma unroll 8
    for (int k = 0; k < kSlots; ++k) {
      scratch[((size_t)k * 2 + 0) * C + c] = 0.f;
      scratch[((size_t)k * 2 + 1) * C + c] = 0.f;
    }
    dsum[c] = s0;
    dsum[C + c] = s1;
    if (dbeta) dbeta[c] += s0;
    if (dgamma) dgamma[c] += s1;
  }
}

template <bool RELU, bool HAS_RES, bool NT>
__global__ __launch_bounds__(kMaxThreads) void bn_apply_kernel(const bf16_t* __restrict__ x,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const bf16_t* __restrict__ res, PadGeom g, int c8,
                                                               int PY, bf16_t* __restrict__ y) {
  const int cx = threadIdx.x % c8, py = threadIdx.x / c8;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cx * 8 + i;
    sc[i] = gamma[c] * rstd[c];
    sh[i] = beta[c] - mean[c] * sc[i];
  }
  for (int p = blockIdx.x * PY + py; p < g.npix; p += gridDim.x * PY) {
    const int o = pad_offset(g, p) + cx * 8;
    float f[8];
    unpack8(ld16_stream<NT>(x + o), f);
    float r[8];
    if (HAS_RES) unpack8(ld16_stream<NT>(res + o), r);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = f[i] * sc[i] + sh[i];
      if (HAS_RES) v += r[i];
      if (RELU) v = v > 0.f ? v : 0.f;
      f[i] = v;
    }
    *(u32x4_t*)(y + o) = pack8(f);
  }
}

// backward pass 1.  POOL: gy comes from gpooled[b][c]/(H*W).  When POOL (or y == nullptr with RELU) the
// relu mask is recomputed as (x*sc + sh > 0) with the exact expression bn_apply_kernel evaluates, so the
// forward output need not be re-read (saves one of three tensor reads per pass).
template <bool RELU, bool POOL>
__global__ __launch_bounds__(kMaxThreads) void bn_bwd_reduce_kernel(const bf16_t* __restrict__ gy,
                                                                    const float* __restrict__ gpooled,
                                                                    const bf16_t* __restrict__ y,
                                                                    const bf16_t* __restrict__ x,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, PadGeom g,
                                                                    int c8, int PY, float* __restrict__ scratch,
                                                                    unsigned slot_mask) {
  extern __shared__ float lds[];
  const int cx = threadIdx.x % c8, py = threadIdx.x / c8;
  const bool maskx = POOL || (RELU && y == nullptr);
  float mu[8], rs[8], sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cx * 8 + i;
    mu[i] = mean[c];
    rs[i] = rstd[c];
    if (maskx) { sc[i] = gamma[c] * rs[i]; sh[i] = beta[c] - mu[i] * sc[i]; }
  }
  const float inv_hw = 1.f / (float)(g.H * g.W);
  const int hw = g.H * g.W;
  float acc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.f;
  auto one = [&](int p, const u32x4_t vx, const u32x4_t vg, const u32x4_t vy) {
    float fx[8], fg[8], fy[8];
    unpack8(vx, fx);
    if (POOL) {
      const int b = p / hw;
#pragma unroll
      for (int i = 0; i < 8; ++i) fg[i] = gpooled[(size_t)b * g.C + cx * 8 + i] * inv_hw;
    } else {
      unpack8(vg, fg);
      if (RELU && !maskx) unpack8(vy, fy);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (fx[i] - mu[i]) * rs[i];
      float gg = fg[i];
      if (maskx) gg = (fx[i] * sc[i] + sh[i]) > 0.f ? gg : 0.f;
      else if (RELU) gg = fy[i] > 0.f ? gg : 0.f;
      acc[0][i] += gg;
      acc[1][i] += gg * xh;
    }
  };
  // two pixels in flight per thread (every load of both issued before the first use): one pixel at a time left the
  // pass at 2.7 TB/s on the whole chip (124 us for 2 x 168 MB)
  const int step = gridDim.x * PY;
  int p = blockIdx.x * PY + py;
  for (; p + step < g.npix; p += 2 * step) {
    const int o0 = pad_offset(g, p) + cx * 8, o1 = pad_offset(g, p + step) + cx * 8;
    const u32x4_t x0 = *(const u32x4_t*)(x + o0), x1 = *(const u32x4_t*)(x + o1);
    u32x4_t g0 = x0, g1 = x1, y0 = x0, y1 = x1;
    if (!POOL) {
      g0 = *(const u32x4_t*)(gy + o0); g1 = *(const u32x4_t*)(gy + o1);
      if (RELU && !maskx) { y0 = *(const u32x4_t*)(y + o0); y1 = *(const u32x4_t*)(y + o1); }
    }
    one(p, x0, g0, y0);
    one(p + step, x1, g1, y1);
  }
  if (p < g.npix) {
    const int o0 = pad_offset(g, p) + cx * 8;
    const u32x4_t x0 = *(const u32x4_t*)(x + o0);
    u32x4_t g0 = x0, y0 = x0;
    if (!POOL) {
      g0 = *(const u32x4_t*)(gy + o0);
      if (RELU && !maskx) y0 = *(const u32x4_t*)(y + o0);
    }
    one(p, x0, g0, y0);
  }
  block_fold_to_slots<2>(acc, cx, py, c8, PY, g.C, scratch, lds, slot_mask);
}

template <bool RELU, bool POOL, bool HAS_ADD, bool HAS_GRES>
__global__ __launch_bounds__(kMaxThreads) void bn_bwd_apply_kernel(
    const bf16_t* __restrict__ gy, const float* __restrict__ gpooled, const bf16_t* __restrict__ y,
    const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ dsum,
    const bf16_t* __restrict__ gx_add, PadGeom g, int c8, int PY, bf16_t* __restrict__ gx,
    bf16_t* __restrict__ g_resid) {
  const int cx = threadIdx.x % c8, py = threadIdx.x / c8;
  const float inv_n = 1.f / (float)g.npix;
  const bool maskx = POOL || (RELU && y == nullptr);
  float mu[8], rs[8], sh[8], k0[8], k1[8], sc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cx * 8 + i;
    mu[i] = mean[c];
    rs[i] = rstd[c];
    sc[i] = gamma[c] * rs[i];
    if (maskx) sh[i] = beta[c] - mu[i] * sc[i];
    k0[i] = dsum[c] * inv_n;
    k1[i] = dsum[g.C + c] * inv_n;
  }
  const float inv_hw = 1.f / (float)(g.H * g.W);
  const int hw = g.H * g.W;
  auto one = [&](int p, int o, const u32x4_t vx, const u32x4_t vg, const u32x4_t vy, const u32x4_t va) {
    float fx[8], fg[8], fy[8], fa[8];
    unpack8(vx, fx);
    if (POOL) {
      const int b = p / hw;
#pragma unroll
      for (int i = 0; i < 8; ++i) fg[i] = gpooled[(size_t)b * g.C + cx * 8 + i] * inv_hw;
    } else {
      unpack8(vg, fg);
      if (RELU && !maskx) unpack8(vy, fy);
    }
    if (HAS_ADD) unpack8(va, fa);
    float out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (fx[i] - mu[i]) * rs[i];
      float gg = fg[i];
      if (maskx) gg = (fx[i] * sc[i] + sh[i]) > 0.f ? gg : 0.f;
      else if (RELU) gg = fy[i] > 0.f ? gg : 0.f;
      fg[i] = gg;
      float v = sc[i] * (gg - k0[i] - xh * k1[i]);
      if (HAS_ADD) v += fa[i];
      out[i] = v;
    }
    *(u32x4_t*)(gx + o) = pack8(out);
    if (HAS_GRES) *(u32x4_t*)(g_resid + o) = pack8(fg);
  };
  auto load = [&](int o, u32x4_t& vx, u32x4_t& vg, u32x4_t& vy, u32x4_t& va) {
    vx = *(const u32x4_t*)(x + o);
    vg = vx; vy = vx; va = vx;
    if (!POOL) {
      vg = *(const u32x4_t*)(gy + o);
      if (RELU && !maskx) vy = *(const u32x4_t*)(y + o);
    }
    if (HAS_ADD) va = *(const u32x4_t*)(gx_add + o);
  };
  // two pixels in flight per thread, like the CU-confined twin below
  const int step = gridDim.x * PY;
  int p = blockIdx.x * PY + py;
  for (; p + step < g.npix; p += 2 * step) {
    const int o0 = pad_offset(g, p) + cx * 8, o1 = pad_offset(g, p + step) + cx * 8;
    u32x4_t x0, g0, y0, a0, x1, g1, y1, a1;
    load(o0, x0, g0, y0, a0);
    load(o1, x1, g1, y1, a1);
    one(p, o0, x0, g0, y0, a0);
    one(p + step, o1, x1, g1, y1, a1);
  }
  if (p < g.npix) {
    const int o0 = pad_offset(g, p) + cx * 8;
    u32x4_t x0, g0, y0, a0;
    load(o0, x0, g0, y0, a0);
    one(p, o0, x0, g0, y0, a0);
  }
}

// The same pass on a FIXED NUMBER OF CUs (nbdt_bn_bwd_apply_cus): `cus` persistent blocks of up to 1024 threads,
// one per CU (the launch asks for 96 KB of LDS it never touches, so a second block cannot join), kCusInFlight pixels in
// flight per thread.  An HBM-bound pass needs few CUs -- 64 reach 3.0 TB/s, 96 4.1, all 256 5.6 (probes/cu_share_probe.hip) --
// and an MFMA-bound kernel loses less than its share of CUs when it gives some up (the chip is power-limited: 192 CUs
// deliver 83 % of the 256-CU matrix rate), so the weight gradient of the same unit runs on the other CUs meanwhile.
// A block of the ordinary launch above is 4 waves and the dispatcher spreads 2048 of them over every CU, where
// they keep an 8-wave weight-gradient block (a CU's whole register file) from starting at all.
// Fused-ReLU form only (mask recomputed from x; what the engines' fused backward uses).
// FOLD (nbdt_bn_bwd_cus): the kernel also does what bn_bwd_finalize_kernel did in a launch of its own between the
// sums and this pass (7 us + a kernel boundary on the backward critical path, 21 times per WRN-28-10 step): every
// block folds the 32 slots of `slots` (ascending slot order, like the finalize kernel: same bits) into its LDS and
// takes k0 / k1 from there; block 0 also writes dsum, accumulates dbeta / dgamma and zeroes `zero_other` -- the
// OTHER slot buffer of the caller's pair, which nobody touches during this launch (the slots being read here cannot
// be zeroed before every block has read them; the caller alternates the two buffers).
#ifndef NBDT_CUS_IN_FLIGHT
#define NBDT_CUS_IN_FLIGHT 2   // 4 measured equal, 8 slower (profiles/r06_session2_small_abs.txt)
#endif
constexpr int kCusInFlight = NBDT_CUS_IN_FLIGHT;   // pixels (16-byte loads per tensor) in flight per thread of the confined passes
template <bool HAS_ADD, bool FOLD, bool NT>
__global__ __launch_bounds__(1024) void bn_bwd_apply_cus_kernel(
    const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ dsum, const bf16_t* __restrict__ gx_add, PadGeom g, int c8, int PY,
    bf16_t* __restrict__ gx, const float* __restrict__ slots, float* __restrict__ zero_other,
    float* __restrict__ dsum_out, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  extern __shared__ float fold_lds[];      // [2][C] (the launch asks for 96 KB to keep a second block off the CU)
  if (FOLD) {
    const int C = g.C;
    for (int j = threadIdx.x; j < 2 * C; j += 1024) {
      const int which = j >= C ? 1 : 0, c = j - which * C;
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < kSlots; ++k) s += slots[((size_t)k * 2 + which) * C + c];
      fold_lds[j] = s;
      if (blockIdx.x == 0) {
        dsum_out[j] = s;
        if (which == 0) { if (dbeta) dbeta[c] += s; }
        else { if (dgamma) dgamma[c] += s; }
      }
    }
    if (blockIdx.x == 0)
      for (int j = threadIdx.x; j < kSlots * 2 * C; j += 1024) zero_other[j] = 0.f;
    __syncthreads();
  }
  const int cx = threadIdx.x % c8, py = threadIdx.x / c8;
  if (py >= PY) return;
  const float inv_n = 1.f / (float)g.npix;
  float mu[8], rs[8], sh[8], k0[8], k1[8], sc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cx * 8 + i;
    mu[i] = mean[c];
    rs[i] = rstd[c];
    sc[i] = gamma[c] * rs[i];
    sh[i] = beta[c] - mu[i] * sc[i];
    k0[i] = (FOLD ? fold_lds[c] : dsum[c]) * inv_n;
    k1[i] = (FOLD ? fold_lds[g.C + c] : dsum[g.C + c]) * inv_n;
  }
  auto one = [&](const u32x4_t vx, const u32x4_t vg, const u32x4_t va, int o) {
    float fx[8], fg[8], fa[8], out[8];
    unpack8(vx, fx);
    unpack8(vg, fg);
    if (HAS_ADD) unpack8(va, fa);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (fx[i] - mu[i]) * rs[i];
      const float gg = (fx[i] * sc[i] + sh[i]) > 0.f ? fg[i] : 0.f;
      float v = sc[i] * (gg - k0[i] - xh * k1[i]);
      if (HAS_ADD) v += fa[i];
      out[i] = v;
    }
    *(u32x4_t*)(gx + o) = pack8(out);
  };
  const int step = gridDim.x * PY;
  int p = blockIdx.x * PY + py;
  for (; p + (kCusInFlight - 1) * step < g.npix; p += kCusInFlight * step) {
    int o[kCusInFlight];
    u32x4_t vx[kCusInFlight], vg[kCusInFlight], va[kCusInFlight];
#pragma unroll
    for (int u = 0; u < kCusInFlight; ++u) o[u] = pad_offset(g, p + u * step) + cx * 8;
#pragma unroll
    for (int u = 0; u < kCusInFlight; ++u) {
      vx[u] = ld16_stream<NT>(x + o[u]);
      vg[u] = ld16_stream<NT>(gy + o[u]);
      va[u] = vx[u];
      if (HAS_ADD) va[u] = ld16_stream<NT>(gx_add + o[u]);
    }
#pragma unroll
    for (int u = 0; u < kCusInFlight; ++u) one(vx[u], vg[u], va[u], o[u]);
  }
  for (; p < g.npix; p += step) {
    const int o0 = pad_offset(g, p) + cx * 8;
    const u32x4_t x0 = ld16_stream<NT>(x + o0), g0 = ld16_stream<NT>(gy + o0);
    u32x4_t a0 = x0;
    if (HAS_ADD) a0 = ld16_stream<NT>(gx_add + o0);
    one(x0, g0, a0, o0);
  }
}

// Reduction twin of bn_bwd_apply_cus_kernel: sum g', sum g'*xhat of relu(bn(x))'s backward on `cus` persistent
// one-per-CU blocks (same thread layout, two pixels in flight), folded per block through the (otherwise only
// occupancy-forcing) dynamic LDS and added to the 32 replicated slots of `scratch` like bn_bwd_reduce_kernel does.
// With it the BatchNorm-backward sums no longer have to come out of the data gradient's epilogue, which reads the
// BatchNorm input in an HBM burst while the matrix pipes wait (+50 us per launch at 32x32x160): the whole
// BatchNorm backward (reduce, fold, apply) becomes HBM-bound work beside the weight gradient.
template <bool NT>
__global__ __launch_bounds__(1024) void bn_bwd_reduce_cus_kernel(const bf16_t* __restrict__ gy,
                                                                 const bf16_t* __restrict__ x,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, PadGeom g, int c8,
                                                                 int PY, float* __restrict__ scratch,
                                                                 unsigned slot_mask) {
  extern __shared__ float lds[];
  const int cx = threadIdx.x % c8, py = threadIdx.x / c8;
  const bool live = py < PY;               // (threads past the last whole pixel row idle, but join the block fold)
  float mu[8], rs[8], sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = (live ? cx : 0) * 8 + i;
    mu[i] = mean[c];
    rs[i] = rstd[c];
    sc[i] = gamma[c] * rs[i];
    sh[i] = beta[c] - mu[i] * sc[i];
  }
  float acc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.f;
  auto one = [&](const u32x4_t vx, const u32x4_t vg) {
    float fx[8], fg[8];
    unpack8(vx, fx);
    unpack8(vg, fg);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (fx[i] - mu[i]) * rs[i];
      const float gg = (fx[i] * sc[i] + sh[i]) > 0.f ? fg[i] : 0.f;
      acc[0][i] += gg;
      acc[1][i] += gg * xh;
    }
  };
  if (live) {
    const int step = gridDim.x * PY;
    int p = blockIdx.x * PY + py;
    for (; p + (kCusInFlight - 1) * step < g.npix; p += kCusInFlight * step) {      // (pixels in ascending order: the sums' bits
      u32x4_t vx[kCusInFlight], vg[kCusInFlight];                                  //  do not depend on kCusInFlight)
#pragma unroll
      for (int u = 0; u < kCusInFlight; ++u) {
        const int o = pad_offset(g, p + u * step) + cx * 8;
        vx[u] = ld16_stream<NT>(x + o);
        vg[u] = ld16_stream<NT>(gy + o);
      }
#pragma unroll
      for (int u = 0; u < kCusInFlight; ++u) one(vx[u], vg[u]);
    }
    for (; p < g.npix; p += step) {
      const int o0 = pad_offset(g, p) + cx * 8;
      one(ld16_stream<NT>(x + o0), ld16_stream<NT>(gy + o0));
    }
  }
  // block fold: [PY][c8][16] floats in LDS, then one atomic per (channel, sum) of the block into its slot
  if (live) {
    float* mine = lds + ((size_t)py * c8 + cx) * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) { mine[i] = acc[0][i]; mine[8 + i] = acc[1][i]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < c8 * 16; e += 1024) {     // e = which * (c8*8) + channel: contiguous adds per wave
    const int which = e / (c8 * 8), c = e - which * (c8 * 8);
    float s = 0.f;
    for (int r = 0; r < PY; ++r) s += lds[((size_t)r * c8 + (c >> 3)) * 16 + which * 8 + (c & 7)];
    atomicAdd(scratch + ((size_t)(blockIdx.x & slot_mask) * 2 + which) * g.C + c, s);
  }
}

// pooled[b][c] = mean_hw relu(bn(x)); one thread per (b, 8-channel chunk)
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const bf16_t* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, PadGeom g,
                                                           float* __restrict__ pooled) {
  const int c8 = g.C / 8;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= g.B * c8) return;
  const int b = idx / c8, cx = idx - b * c8;
  float sc[8], sh[8], acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cx * 8 + i;
    sc[i] = gamma[c] * rstd[c];
    sh[i] = beta[c] - mean[c] * sc[i];
    acc[i] = 0.f;
  }
  for (int h = 0; h < g.H; ++h)
    for (int w = 0; w < g.W; ++w) {
      float f[8];
      unpack8(*(const u32x4_t*)(x + (size_t)b * g.img + (h + 1) * g.row + (w + 1) * g.C + cx * 8), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = f[i] * sc[i] + sh[i];
        acc[i] += v > 0.f ? v : 0.f;
      }
    }
  const float inv = 1.f / (float)(g.H * g.W);
#pragma unroll
  for (int i = 0; i < 8; ++i) pooled[(size_t)b * g.C + cx * 8 + i] = acc[i] * inv;
}

// ------------------------------------------------------------------------------------------ host

static int check_shape(int B, int H, int W, int C) {
  NBDT_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "empty tensor");
  NBDT_REQUIRE(C % 8 == 0 && C / 8 <= kMaxThreads, "C must be a multiple of 8 and <= 2048");
  NBDT_REQUIRE((long long)B * (H + 2) * (W + 2) * C < (1ll << 31), "tensor too large for 32-bit offsets");
  return NBDT_OK;
}


// Where a reduction kernel of `grid` blocks adds its per-block sums of n = 2*C floats.  Normally the caller's 32-slot
// scratch; in deterministic mode (nbdt_set_deterministic) a zeroed library-owned row per block -- each address then
// receives exactly one atomic add -- that slot_finish() sums in block order into slot 0 of the caller's scratch, so
// the finalize kernels (which fold the 32 slots in a fixed order anyway) need no second form.
struct SlotTarget {
  float* ptr;
  unsigned mask;
  bool det;
};
static int slot_target(hipStream_t st, float* scratch, int grid, size_t n, SlotTarget* t) {
  t->ptr = scratch; t->mask = kSlots - 1; t->det = false;
  if (!deterministic()) return NBDT_OK;
  float* rows = det_rows(st, (size_t)grid * n);
  if (!rows) return nbdt::fail(NBDT_ENOMEM, "deterministic mode: %s (%s)", "no workspace for the per-block rows", nbdt::det_rows_why());
  NBDT_HIP_CHECK(hipMemsetAsync(rows, 0, (size_t)grid * n * sizeof(float), st));
  t->ptr = rows; t->mask = ~0u; t->det = true;
  return NBDT_OK;
}
static int slot_finish(hipStream_t st, const SlotTarget& t, int grid, size_t n, float* scratch) {
  return t.det ? det_fold(st, t.ptr, grid, n, scratch) : NBDT_OK;
}

extern "C" int nbdt_bn_stats(const void* x, int32_t B, int32_t H, int32_t W, int32_t C, float eps, float momentum,
                             float* running_mean, float* running_var, float* scratch, float* save_mean,
                             float* save_rstd, void* stream) {
  NBDT_REQUIRE(scratch && save_mean && save_rstd, "null argument");
  NBDT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running stats must be both set or both NULL");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  if (x != nullptr) {   // x == NULL: the producer (nbdt_dwconv_fwd with bn_scratch) already filled the slots
    const Layout l = layout_for(C);
    const size_t shmem = (size_t)l.threads * 16 * sizeof(float);
    const int grid = grid_for(g, l, 16);
    SlotTarget t;
    rc = slot_target(st, scratch, grid, 2 * (size_t)C, &t);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(grid), dim3(l.threads), shmem, st, (const bf16_t*)x, g,
                       l.c8, l.py, t.ptr, t.mask);
    NBDT_LAUNCH_CHECK();
    rc = slot_finish(st, t, grid, 2 * (size_t)C, scratch);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, C, (float)g.npix, eps, momentum,
                     running_mean, running_var, save_mean, save_rstd);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

// fold [rows][2][C] partial sums written by conv epilogues: block = CL channels x (1024 / CL) row lanes, every
// thread keeps 4 independent loads in flight (a 5-block, 8-row-lane version took 39 us at 2048 rows).
// CL = 32: 32 row lanes (short tables).  CL = 8: 128 row lanes and four times the blocks for tall tables (2048 rows of
// a 32x32x160 layer = 2.6 MB).  Round 4 measured what this kernel's 11-13 us after such a conv are: NOT its structure
// (5 or 20 or 40 blocks, 8 or 16 loads in flight, serial or butterfly row reduction: all the same) but the kernel
// boundary writing back the 168 MB the conv left dirty in the L2s -- whichever kernel comes next pays it.
// Same summation tree for a given (rows, CL): deterministic.
template <int CL>
__global__ __launch_bounds__(1024) void bn_fold_partials_kernel(const float* __restrict__ part, int rows, int C,
                                                                float n, float eps, float momentum,
                                                                float* __restrict__ running_mean,
                                                                float* __restrict__ running_var,
                                                                float* __restrict__ save_mean,
                                                                float* __restrict__ save_rstd) {
  constexpr int RL = 1024 / CL;
  static_assert(CL <= 32 && 64 % CL == 0, "a wave holds whole groups of CL channels");
  __shared__ float red[2][16][CL + 1];
  const int cl = threadIdx.x % CL, rs = threadIdx.x / CL;
  const int c = blockIdx.x * CL + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
  if (c < C) {
    int r = rs;
    for (; r + 3 * RL < rows; r += 4 * RL) {
      s0 += part[((size_t)r * 2 + 0) * C + c];             q0 += part[((size_t)r * 2 + 1) * C + c];
      s1 += part[((size_t)(r + RL) * 2 + 0) * C + c];      q1 += part[((size_t)(r + RL) * 2 + 1) * C + c];
      s2 += part[((size_t)(r + 2 * RL) * 2 + 0) * C + c];  q2 += part[((size_t)(r + 2 * RL) * 2 + 1) * C + c];
      s3 += part[((size_t)(r + 3 * RL) * 2 + 0) * C + c];  q3 += part[((size_t)(r + 3 * RL) * 2 + 1) * C + c];
    }
    for (; r < rows; r += RL) {
      s0 += part[((size_t)r * 2 + 0) * C + c];
      q0 += part[((size_t)r * 2 + 1) * C + c];
    }
  }
  // row lanes -> one value per channel: butterfly over the lanes of a wave that hold the same channel (lane bits
  // >= log2 CL), then the 16 waves' values through LDS
  float ws = (s0 + s1) + (s2 + s3), wq = (q0 + q1) + (q2 + q3);
#pragma unroll
  for (int off = CL; off < 64; off <<= 1) {
    ws += __shfl_xor(ws, off, 64);
    wq += __shfl_xor(wq, off, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < CL) { red[0][wave][lane] = ws; red[1][wave][lane] = wq; }
  __syncthreads();
  if (rs == 0 && c < C) {      // (rs == 0: threads 0 .. CL-1, cl == threadIdx.x)
    float s = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { s += red[0][k][cl]; sq += red[1][k][cl]; }
    const float mean = s / n;
    float var = sq / n - mean * mean;
    var = var > 0.f ? var : 0.f;
    save_mean[c] = mean;
    save_rstd[c] = rsqrtf(var + eps);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      const float unbiased = n > 1.f ? var * n / (n - 1.f) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
}

// backward twin of bn_fold_partials_kernel: rows of {sum g', sum g'*xhat} -> dsum (+= dbeta / dgamma)
__global__ __launch_bounds__(1024) void bn_bwd_fold_partials_kernel(const float* __restrict__ part, int rows, int C,
                                                                    float* __restrict__ dsum,
                                                                    float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta) {
  __shared__ float red[2][32][33];
  const int cl = threadIdx.x & 31, rs = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
  if (c < C) {
    int r = rs;
    for (; r + 32 < rows; r += 64) {
      s0 += part[((size_t)r * 2 + 0) * C + c];         q0 += part[((size_t)r * 2 + 1) * C + c];
      s1 += part[((size_t)(r + 32) * 2 + 0) * C + c];  q1 += part[((size_t)(r + 32) * 2 + 1) * C + c];
    }
    for (; r < rows; r += 32) {
      s0 += part[((size_t)r * 2 + 0) * C + c];
      q0 += part[((size_t)r * 2 + 1) * C + c];
    }
  }
  red[0][rs][cl] = s0 + s1;
  red[1][rs][cl] = q0 + q1;
  __syncthreads();
  if (rs == 0 && c < C) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < 32; ++k) { a += red[0][k][cl]; b += red[1][k][cl]; }
    dsum[c] = a;
    dsum[C + c] = b;
    if (dbeta) dbeta[c] += a;
    if (dgamma) dgamma[c] += b;
  }
}

extern "C" int nbdt_bn_bwd_fold(int32_t B, int32_t H, int32_t W, int32_t C, const float* bn_partials, float* dsum,
                                float* dgamma, float* dbeta, void* stream) {
  NBDT_REQUIRE(bn_partials && dsum, "null argument");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  const int rows = (int)(((long long)B * H * W + 255) / 256);
  hipLaunchKernelGGL(bn_bwd_fold_partials_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, bn_partials,
                     rows, C, dsum, dgamma, dbeta);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_bn_finalize(int32_t B, int32_t H, int32_t W, int32_t C, float eps, float momentum,
                                float* running_mean, float* running_var, const float* partials, float* save_mean,
                                float* save_rstd, void* stream) {
  NBDT_REQUIRE(partials && save_mean && save_rstd, "null argument");
  NBDT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running stats must be both set or both NULL");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  const long long npix = (long long)B * H * W;
  const int rows = (int)((npix + 255) / 256);   // = the pixel tiles of nbdt_conv_igemm_stats
  if (rows >= 1024)     // tall table: four times the blocks (see the kernel header)
    hipLaunchKernelGGL(bn_fold_partials_kernel<8>, dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)stream, partials, rows,
                       C, (float)npix, eps, momentum, running_mean, running_var, save_mean, save_rstd);
  else
    hipLaunchKernelGGL(bn_fold_partials_kernel<32>, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, partials,
                       rows, C, (float)npix, eps, momentum, running_mean, running_var, save_mean, save_rstd);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_bn_apply(const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                             const float* beta, const void* residual, int32_t relu, int32_t B, int32_t H, int32_t W,
                             int32_t C, void* y, void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && y, "null argument");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  const Layout l = layout_for(C);
  const dim3 grid(grid_for(g, l, 8)), blk(l.threads);
  const bool nt = stream_nt((long long)B * g.img * 2);      // (nontemporal loads: tensors that cannot stay in the Infinity Cache)
#define NBDT_APPLY(R, S)                                                                                         \
  do {                                                                                                           \
    if (nt) hipLaunchKernelGGL((bn_apply_kernel<R, S, true>), grid, blk, 0, st, (const bf16_t*)x, save_mean, save_rstd, gamma, \
                               beta, (const bf16_t*)residual, g, l.c8, l.py, (bf16_t*)y);                          \
    else hipLaunchKernelGGL((bn_apply_kernel<R, S, false>), grid, blk, 0, st, (const bf16_t*)x, save_mean, save_rstd, gamma, \
                            beta, (const bf16_t*)residual, g, l.c8, l.py, (bf16_t*)y);                             \
  } while (0)
  if (relu) { if (residual) NBDT_APPLY(true, true); else NBDT_APPLY(true, false); }
  else { if (residual) NBDT_APPLY(false, true); else NBDT_APPLY(false, false); }
#undef NBDT_APPLY
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

// y = [relu](bn(x)) written as the SPACE-TO-DEPTH copy the stride-2 convolutions of a shape-changing unit read
// (csrc/conv_seg.hip): y is [B][H/2+2][W/2+2][4C]; input pixel (h, w) lands at pixel (h/2, w/2), channels
// [((h&1)*2 + (w&1)) * C, +C).  Same arithmetic as bn_apply_kernel (the ReLU-mask recomputation of the backward
// passes agrees), same bytes; nobody reads the activated tensor of such a unit in its plain layout.
template <bool RELU>
__global__ __launch_bounds__(kMaxThreads) void bn_apply_s2d_kernel(const bf16_t* __restrict__ x,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, PadGeom g, int c8,
                                                                   int PY, bf16_t* __restrict__ y) {
  const int cx = threadIdx.x % c8, py = threadIdx.x / c8;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cx * 8 + i;
    sc[i] = gamma[c] * rstd[c];
    sh[i] = beta[c] - mean[c] * sc[i];
  }
  const int row2 = (g.W / 2 + 2) * 4 * g.C, img2 = (g.H / 2 + 2) * row2;
  for (int p = blockIdx.x * PY + py; p < g.npix; p += gridDim.x * PY) {
    const unsigned t = fdiv((unsigned)p, g.div_w);
    const int w = p - (int)t * g.W;
    const unsigned b = fdiv(t, g.div_h);
    const int h = (int)t - (int)b * g.H;
    const int o = (int)b * g.img + (h + 1) * g.row + (w + 1) * g.C + cx * 8;
    const int o2 = (int)b * img2 + ((h >> 1) + 1) * row2 + ((w >> 1) + 1) * 4 * g.C + ((h & 1) * 2 + (w & 1)) * g.C + cx * 8;
    float f[8];
    unpack8(*(const u32x4_t*)(x + o), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = f[i] * sc[i] + sh[i];
      if (RELU) v = v > 0.f ? v : 0.f;
      f[i] = v;
    }
    *(u32x4_t*)(y + o2) = pack8(f);
  }
}

extern "C" int nbdt_bn_apply_s2d(const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                                 const float* beta, int32_t relu, int32_t B, int32_t H, int32_t W, int32_t C, void* y,
                                 void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && y, "null argument");
  NBDT_REQUIRE(H % 2 == 0 && W % 2 == 0, "space-to-depth needs even H and W");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  const Layout l = layout_for(C);
  const dim3 grid(grid_for(g, l, 8)), blk(l.threads);
  if (relu)
    hipLaunchKernelGGL((bn_apply_s2d_kernel<true>), grid, blk, 0, st, (const bf16_t*)x, save_mean, save_rstd, gamma,
                       beta, g, l.c8, l.py, (bf16_t*)y);
  else
    hipLaunchKernelGGL((bn_apply_s2d_kernel<false>), grid, blk, 0, st, (const bf16_t*)x, save_mean, save_rstd, gamma,
                       beta, g, l.c8, l.py, (bf16_t*)y);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_bn_bwd_reduce(const void* gy, const void* y, const void* x, const float* save_mean,
                                  const float* save_rstd, const float* gamma, const float* beta, int32_t relu,
                                  int32_t B, int32_t H, int32_t W, int32_t C, float* scratch, float* dsum,
                                  float* dgamma, float* dbeta, void* stream) {
  NBDT_REQUIRE(gy && x && save_mean && save_rstd && scratch && dsum, "null argument");
  NBDT_REQUIRE(!relu || y || (gamma && beta), "relu backward needs y, or gamma+beta to recompute the mask");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  const Layout l = layout_for(C);
  const size_t shmem = (size_t)l.threads * 16 * sizeof(float);
  const int nblk = grid_for(g, l, 16);
  const dim3 grid(nblk), blk(l.threads);
  SlotTarget t;
  rc = slot_target(st, scratch, nblk, 2 * (size_t)C, &t);
  if (rc) return rc;
  if (relu)
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, false>), grid, blk, shmem, st, (const bf16_t*)gy, nullptr,
                       (const bf16_t*)y, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, l.c8, l.py,
                       t.ptr, t.mask);
  else
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<false, false>), grid, blk, shmem, st, (const bf16_t*)gy, nullptr,
                       (const bf16_t*)y, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, l.c8, l.py,
                       t.ptr, t.mask);
  NBDT_LAUNCH_CHECK();
  rc = slot_finish(st, t, nblk, 2 * (size_t)C, scratch);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, C, dsum, dgamma, dbeta);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_bn_bwd_apply(const void* gy, const void* y, const void* x, const float* save_mean,
                                 const float* save_rstd, const float* gamma, const float* beta, const float* dsum,
                                 const void* gx_add, int32_t relu, int32_t B, int32_t H, int32_t W, int32_t C,
                                 void* gx, void* g_resid, void* stream) {
  NBDT_REQUIRE(gy && x && save_mean && save_rstd && gamma && dsum && gx, "null argument");
  NBDT_REQUIRE(!relu || y || beta, "relu backward needs y, or beta to recompute the mask");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  const Layout l = layout_for(C);
  const dim3 grid(grid_for(g, l, 8)), blk(l.threads);
#define NBDT_BA(R, A, G)                                                                                          \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<R, false, A, G>), grid, blk, 0, st, (const bf16_t*)gy, nullptr,          \
                     (const bf16_t*)y, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, dsum,                  \
                     (const bf16_t*)gx_add, g, l.c8, l.py, (bf16_t*)gx, (bf16_t*)g_resid)
  const bool a = gx_add != nullptr, r = g_resid != nullptr;
  if (relu) {
    if (a) { if (r) NBDT_BA(true, true, true); else NBDT_BA(true, true, false); }
    else { if (r) NBDT_BA(true, false, true); else NBDT_BA(true, false, false); }
  } else {
    if (a) { if (r) NBDT_BA(false, true, true); else NBDT_BA(false, true, false); }
    else { if (r) NBDT_BA(false, false, true); else NBDT_BA(false, false, false); }
  }
#undef NBDT_BA
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_bn_bwd_apply_cus(const void* gy, const void* x, const float* save_mean, const float* save_rstd,
                                    const float* gamma, const float* beta, const float* dsum, const void* gx_add,
                                    int32_t B, int32_t H, int32_t W, int32_t C, void* gx, int32_t cus,
                                    void* stream) {
  NBDT_REQUIRE(gy && x && save_mean && save_rstd && gamma && beta && dsum && gx, "null argument");
  NBDT_REQUIRE(cus >= 1 && cus <= 256, "cus must be 1..256");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  const PadGeom g = make_geom(B, H, W, C);
  const int c8 = C / 8;
  const int py = 1024 / c8;                       // C <= 2048 (check_shape): at least 4 pixel rows
  const bool nt = stream_nt((long long)B * g.img * 2);
  constexpr int kForceLds = 96 * 1024;            // more than half a CU's LDS: one block per CU
  static DeviceAttr site;
  if (site.need(kForceLds)) {
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_apply_cus_kernel<true, false, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_apply_cus_kernel<true, false, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_apply_cus_kernel<false, false, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_apply_cus_kernel<false, false, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    site.done(kForceLds);
  }
  int blocks = cus;
  const int max_blocks = (g.npix + py - 1) / py;
  if (blocks > max_blocks) blocks = max_blocks;
  const dim3 grid(blocks), blk(1024);
  if (gx_add)
    { if (nt) hipLaunchKernelGGL((bn_bwd_apply_cus_kernel<true, false, true>), grid, blk, kForceLds, (hipStream_t)stream,
                       (const bf16_t*)gy, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, dsum,
                       (const bf16_t*)gx_add, g, c8, py, (bf16_t*)gx, nullptr, nullptr, nullptr, nullptr, nullptr);
      else hipLaunchKernelGGL((bn_bwd_apply_cus_kernel<true, false, false>), grid, blk, kForceLds, (hipStream_t)stream,
                       (const bf16_t*)gy, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, dsum,
                       (const bf16_t*)gx_add, g, c8, py, (bf16_t*)gx, nullptr, nullptr, nullptr, nullptr, nullptr); }
  else
    { if (nt) hipLaunchKernelGGL((bn_bwd_apply_cus_kernel<false, false, true>), grid, blk, kForceLds, (hipStream_t)stream,
                       (const bf16_t*)gy, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, dsum, nullptr, g, c8, py,
                       (bf16_t*)gx, nullptr, nullptr, nullptr, nullptr, nullptr);
      else hipLaunchKernelGGL((bn_bwd_apply_cus_kernel<false, false, false>), grid, blk, kForceLds, (hipStream_t)stream,
                       (const bf16_t*)gy, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, dsum, nullptr, g, c8, py,
                       (bf16_t*)gx, nullptr, nullptr, nullptr, nullptr, nullptr); }
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_bn_bwd_reduce_cus(const void* gy, const void* x, const float* save_mean, const float* save_rstd,
                                     const float* gamma, const float* beta, int32_t B, int32_t H, int32_t W,
                                     int32_t C, float* scratch, float* dsum, float* dgamma, float* dbeta,
                                     int32_t cus, void* stream) {
  NBDT_REQUIRE(gy && x && save_mean && save_rstd && gamma && beta && scratch && dsum, "null argument");
  NBDT_REQUIRE(cus >= 1 && cus <= 256, "cus must be 1..256");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  const int c8 = C / 8;
  const int py = 1024 / c8;
  const bool nt = stream_nt((long long)B * g.img * 2);
  constexpr int kForceLds = 96 * 1024;            // one block per CU; the block fold uses py*c8*64 <= 64 KB of it
  static DeviceAttr site;
  if (site.need(kForceLds)) {
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_reduce_cus_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_reduce_cus_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    site.done(kForceLds);
  }
  int blocks = cus;
  const int max_blocks = (g.npix + py - 1) / py;
  if (blocks > max_blocks) blocks = max_blocks;
  SlotTarget t;
  rc = slot_target(st, scratch, blocks, 2 * (size_t)C, &t);
  if (rc) return rc;
  { if (nt) hipLaunchKernelGGL((bn_bwd_reduce_cus_kernel<true>), dim3(blocks), dim3(1024), kForceLds, st, (const bf16_t*)gy,
                     (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, c8, py, t.ptr, t.mask);
    else hipLaunchKernelGGL((bn_bwd_reduce_cus_kernel<false>), dim3(blocks), dim3(1024), kForceLds, st, (const bf16_t*)gy,
                     (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, c8, py, t.ptr, t.mask); }
  NBDT_LAUNCH_CHECK();
  rc = slot_finish(st, t, blocks, 2 * (size_t)C, scratch);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, C, dsum, dgamma, dbeta);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_bn_bwd_cus(const void* gy, const void* x, const float* save_mean, const float* save_rstd,
                               const float* gamma, const float* beta, const void* gx_add, int32_t B, int32_t H,
                               int32_t W, int32_t C, float* slots, float* slots_other, float* dsum, float* dgamma,
                               float* dbeta, void* gx, int32_t cus, void* stream) {
  NBDT_REQUIRE(gy && x && save_mean && save_rstd && gamma && beta && slots && slots_other && dsum && gx, "null argument");
  NBDT_REQUIRE(slots != slots_other, "the two slot buffers must be distinct");
  NBDT_REQUIRE(cus >= 1 && cus <= 256, "cus must be 1..256");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  const int c8 = C / 8;
  const int py = 1024 / c8;
  const bool nt = stream_nt((long long)B * g.img * 2);
  constexpr int kForceLds = 96 * 1024;
  static DeviceAttr site;
  if (site.need(kForceLds)) {
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_reduce_cus_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_reduce_cus_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_apply_cus_kernel<true, true, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_apply_cus_kernel<true, true, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_apply_cus_kernel<false, true, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    NBDT_ATTR_CHECK(site, hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_apply_cus_kernel<false, true, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kForceLds));
    site.done(kForceLds);
  }
  int blocks = cus;
  const int max_blocks = (g.npix + py - 1) / py;
  if (blocks > max_blocks) blocks = max_blocks;
  SlotTarget t;
  rc = slot_target(st, slots, blocks, 2 * (size_t)C, &t);
  if (rc) return rc;
  { if (nt) hipLaunchKernelGGL((bn_bwd_reduce_cus_kernel<true>), dim3(blocks), dim3(1024), kForceLds, st, (const bf16_t*)gy,
                     (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, c8, py, t.ptr, t.mask);
    else hipLaunchKernelGGL((bn_bwd_reduce_cus_kernel<false>), dim3(blocks), dim3(1024), kForceLds, st, (const bf16_t*)gy,
                     (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, c8, py, t.ptr, t.mask); }
  NBDT_LAUNCH_CHECK();
  rc = slot_finish(st, t, blocks, 2 * (size_t)C, slots);      // (deterministic mode: the block rows -> slot 0)
  if (rc) return rc;
  if (gx_add)
    { if (nt) hipLaunchKernelGGL((bn_bwd_apply_cus_kernel<true, true, true>), dim3(blocks), dim3(1024), kForceLds, st, (const bf16_t*)gy,
                       (const bf16_t*)x, save_mean, save_rstd, gamma, beta, nullptr, (const bf16_t*)gx_add, g, c8, py,
                       (bf16_t*)gx, slots, slots_other, dsum, dgamma, dbeta);
      else hipLaunchKernelGGL((bn_bwd_apply_cus_kernel<true, true, false>), dim3(blocks), dim3(1024), kForceLds, st, (const bf16_t*)gy,
                       (const bf16_t*)x, save_mean, save_rstd, gamma, beta, nullptr, (const bf16_t*)gx_add, g, c8, py,
                       (bf16_t*)gx, slots, slots_other, dsum, dgamma, dbeta); }
  else
    { if (nt) hipLaunchKernelGGL((bn_bwd_apply_cus_kernel<false, true, true>), dim3(blocks), dim3(1024), kForceLds, st,
                       (const bf16_t*)gy, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, nullptr, nullptr, g, c8, py,
                       (bf16_t*)gx, slots, slots_other, dsum, dgamma, dbeta);
      else hipLaunchKernelGGL((bn_bwd_apply_cus_kernel<false, true, false>), dim3(blocks), dim3(1024), kForceLds, st,
                       (const bf16_t*)gy, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, nullptr, nullptr, g, c8, py,
                       (bf16_t*)gx, slots, slots_other, dsum, dgamma, dbeta); }
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_bn_relu_pool(const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                                 const float* beta, int32_t B, int32_t H, int32_t W, int32_t C, float* pooled,
                                 void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && pooled, "null argument");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  const PadGeom g = make_geom(B, H, W, C);
  const int n = B * (C / 8);
  hipLaunchKernelGGL(bn_relu_pool_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, pooled);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_pool_bn_bwd_reduce(const float* gpooled, const void* x, const float* save_mean,
                                       const float* save_rstd, const float* gamma, const float* beta, int32_t B,
                                       int32_t H, int32_t W, int32_t C, float* scratch, float* dsum, float* dgamma,
                                       float* dbeta, void* stream) {
  NBDT_REQUIRE(gpooled && x && save_mean && save_rstd && gamma && beta && scratch && dsum, "null argument");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  const Layout l = layout_for(C);
  const size_t shmem = (size_t)l.threads * 16 * sizeof(float);
  const int nblk = grid_for(g, l, 16);
  SlotTarget t;
  rc = slot_target(st, scratch, nblk, 2 * (size_t)C, &t);
  if (rc) return rc;
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, true>), dim3(nblk), dim3(l.threads), shmem, st,
                     nullptr, gpooled, nullptr, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, l.c8, l.py,
                     t.ptr, t.mask);
  NBDT_LAUNCH_CHECK();
  rc = slot_finish(st, t, nblk, 2 * (size_t)C, scratch);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, C, dsum, dgamma, dbeta);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_pool_bn_bwd_apply(const float* gpooled, const void* x, const float* save_mean,
                                      const float* save_rstd, const float* gamma, const float* beta,
                                      const float* dsum, int32_t B, int32_t H, int32_t W, int32_t C, void* gx,
                                      void* stream) {
  NBDT_REQUIRE(gpooled && x && save_mean && save_rstd && gamma && beta && dsum && gx, "null argument");
  int rc = check_shape(B, H, W, C);
  if (rc) return rc;
  const PadGeom g = make_geom(B, H, W, C);
  const Layout l = layout_for(C);
  hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true, false, false>), dim3(grid_for(g, l, 8)), dim3(l.threads), 0,
                     (hipStream_t)stream, nullptr, gpooled, nullptr, (const bf16_t*)x, save_mean, save_rstd, gamma,
                     beta, dsum, nullptr, g, l.c8, l.py, (bf16_t*)gx, nullptr);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}
