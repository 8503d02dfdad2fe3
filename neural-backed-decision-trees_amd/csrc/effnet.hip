// MBConv building blocks of EfficientNet-B0 on gfx950 (pytorchcv `efficientnet_b0`
// behind reference nbdt/models/__init__.py:3): depthwise 3x3/5x5 convolution, BatchNorm + swish,
// squeeze-and-excitation, dropout.  The 1x1 expand/project convolutions are the implicit-GEMM kernels
// of conv_dma.hip; everything here is an HBM-bound streaming pass over padded NHWC bf16 tensors
// (16-byte / 8-channel vectors per lane, a thread keeps ONE channel chunk and walks the pixels of ONE
// image, so per-channel BatchNorm parameters and the per-(image, channel) SE gate live in registers).
//
// Nothing is materialised that can be recomputed from the raw conv output x in the same pass:
//   a = swish(bn(x)) is never stored for the SE branch -- the pool pass reads x only, the scale pass
//   writes a*gate directly, and every backward pass recomputes a / swish'(bn(x)) from x.
// Algorithmic bytes per element: pool 2, apply/scale 4 (+2 residual), se_bwd_reduce 4,
// bwd_reduce 4, bwd_apply 6 (+2 gx_add); depthwise fwd 4, bwd_data 4, bwd_weight 4.
#include "common.h"
#include <stdlib.h>

using namespace nbdt;

namespace {

constexpr int kSlots = NBDT_BN_SLOTS;
constexpr int kThreads = 256;

struct MbGeom {
  int B, H, W, C;
  int hw, row, img;
  FastDiv div_w;
  int c8, PY, threads, slices;
};

MbGeom mb_geom(int B, int H, int W, int C, int ppt) {
  MbGeom g;
  g.B = B; g.H = H; g.W = W; g.C = C;
  g.hw = H * W;
  g.row = (W + 2) * C;
  g.img = (H + 2) * g.row;
  g.div_w = make_fastdiv((unsigned)W);
  g.c8 = C / 8;
  g.PY = kThreads / g.c8;
  if (g.PY < 1) g.PY = 1;
  if (g.PY > g.hw) g.PY = g.hw;
  g.threads = g.c8 * g.PY;
  g.slices = (g.hw + g.PY * ppt - 1) / (g.PY * ppt);
  if (g.slices < 1) g.slices = 1;
  return g;
}

int check_mb(int B, int H, int W, int C) {
  NBDT_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "empty tensor");
  NBDT_REQUIRE(C % 8 == 0 && C / 8 <= kThreads, "C must be a multiple of 8 and <= 2048");
  NBDT_REQUIRE((long long)B * (H + 2) * (W + 2) * C < (1ll << 31), "tensor too large for 32-bit offsets");
  return NBDT_OK;
}

__device__ __forceinline__ int pix_off(const MbGeom& g, int b, int p, int cx) {
  const int h = (int)fdiv((unsigned)p, g.div_w);
  const int w = p - h * g.W;
  return b * g.img + (h + 1) * g.row + (w + 1) * g.C + cx * 8;
}

// Thread blocks go to the 8 XCDs round-robin by linear index (MI355X_MICROARCH.md: block b runs on XCD b % 8), each
// with its own L2: linear neighbours -- adjacent rows of one image in the depthwise kernels, which share K - 1 of their
// K input rows -- never share a cache, and every shared row crossed the fabric once per reader (rocprofv3 FETCH_SIZE:
// 2.4-3.7x the input bytes per depthwise launch, profiles/r06_c5_traffic_by_kernel_before.txt).  This maps XCD i's
// blocks (linear index i, i + 8, ...) to ONE contiguous logical range, so neighbours in the logical order run on the
// same XCD at about the same time.  A bijection on [0, total) for any total; only speed depends on the dispatch order.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned lin, unsigned total) {
#ifdef NBDT_NO_XCD_CONTIGUOUS      // timing-only builds (scratch/build_variants.sh): the dispatch order of rounds 2-5
  return lin;
#endif
  const unsigned q = total >> 3, r = total & 7u, x = lin & 7u, k = lin >> 3;
  return x * q + (x < r ? x : r) + k;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

template <int ACT>
__device__ __forceinline__ float act_fwd(float v) {
  if (ACT == NBDT_ACT_RELU) return v > 0.f ? v : 0.f;
  if (ACT == NBDT_ACT_SWISH) return v * sigmoidf_(v);
  return v;
}
template <int ACT>
__device__ __forceinline__ float act_bwd(float v) {  // d act / d v
  if (ACT == NBDT_ACT_RELU) return v > 0.f ? 1.f : 0.f;
  if (ACT == NBDT_ACT_SWISH) {
    const float s = sigmoidf_(v);
    return s * (1.f + v * (1.f - s));
  }
  return 1.f;
}

struct BnRegs {
  float sc[8], sh[8];
};
__device__ __forceinline__ void load_bn(BnRegs& r, const float* mean, const float* rstd, const float* gamma,
                                        const float* beta, int cx) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cx * 8 + i;
    r.sc[i] = gamma[c] * rstd[c];
    r.sh[i] = beta[c] - mean[c] * r.sc[i];
  }
}

// fold NQ*8 per-thread partials over the PY pixel rows of the block and hand every sum to `sink(q, channel, value)`.
// Output o = q * (c8*8) + channel goes to thread o % (c8*PY): CONSECUTIVE THREADS HOLD CONSECUTIVE CHANNELS, so the
// sinks' global atomics are one contiguous 256-byte run per wave.  (Round 1 gave chunk cx's element e to thread
// (cx, e % PY): lanes 32 bytes apart, 64 sectors per atomic instruction -- that, not the arithmetic, was most of the
// time of every statistics / weight-gradient kernel of this file on the narrow late layers: profiles/r04_dw_wgrad.txt.)
// A sum is still its PY partials in ascending row order: the same bits as before.
template <int NQ, typename Sink>
__device__ __forceinline__ void block_fold(float (&acc)[NQ][8], int cx, int py, int c8, int PY, float* lds,
                                           Sink sink) {
  float* mine = lds + ((size_t)py * c8 + cx) * (NQ * 8);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int i = 0; i < 8; ++i) mine[q * 8 + i] = acc[q][i];
  __syncthreads();
  const int nch = c8 * 8, nthr = c8 * PY;
  for (int o = py * c8 + cx; o < NQ * nch; o += nthr) {
    const int q = o / nch, c = o - q * nch;
    const float* src = lds + (size_t)(c >> 3) * (NQ * 8) + q * 8 + (c & 7);
    float s = 0.f;
    for (int r = 0; r < PY; ++r) s += src[(size_t)r * c8 * (NQ * 8)];
    sink(q, c, s);
  }
}

// ------------------------------------------------------------------------------------------
// y = act(bn(x)) [* gate[b,c]] [+ residual]
template <int ACT, bool GATE, bool RES>
__global__ __launch_bounds__(kThreads) void mb_apply_kernel(const bf16_t* __restrict__ x, const float* mean,
                                                            const float* rstd, const float* gamma,
                                                            const float* beta, const float* __restrict__ gate,
                                                            const bf16_t* __restrict__ res, MbGeom g, int ppt,
                                                            bf16_t* __restrict__ y) {
  const int cx = threadIdx.x % g.c8, py = threadIdx.x / g.c8, b = blockIdx.y;
  BnRegs bn;
  load_bn(bn, mean, rstd, gamma, beta, cx);
  float gt[8];
  if (GATE)
#pragma unroll
    for (int i = 0; i < 8; ++i) gt[i] = gate[(size_t)b * g.C + cx * 8 + i];
  const int p0 = blockIdx.x * g.PY * ppt + py;
  for (int k = 0; k < ppt; ++k) {
    const int p = p0 + k * g.PY;
    if (p >= g.hw) break;
    const int o = pix_off(g, b, p, cx);
    float f[8], r[8];
    unpack8(*(const u32x4_t*)(x + o), f);
    if (RES) unpack8(*(const u32x4_t*)(res + o), r);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = act_fwd<ACT>(f[i] * bn.sc[i] + bn.sh[i]);
      if (GATE) v *= gt[i];
      if (RES) v += r[i];
      f[i] = v;
    }
    *(u32x4_t*)(y + o) = pack8(f);
  }
}

// out[b][c] += scale * sum_pixels  act(bn(x)) [* gu]      (SE squeeze; SE backward dL/dgate)
template <int ACT, bool MUL>
__global__ __launch_bounds__(kThreads) void mb_pool_kernel(const bf16_t* __restrict__ x, const float* mean,
                                                           const float* rstd, const float* gamma,
                                                           const float* beta, const bf16_t* __restrict__ gu,
                                                           MbGeom g, int ppt, float scale,
                                                           float* __restrict__ out, long long row_stride) {
  // row_stride: 0 = every pixel slice of an image adds into out[b]; deterministic mode: B*C, a zeroed copy of `out`
  // per slice (one add per address; det_fold sums the copies in slice order)
  extern __shared__ float lds[];
  const int cx = threadIdx.x % g.c8, py = threadIdx.x / g.c8, b = blockIdx.y;
  BnRegs bn;
  load_bn(bn, mean, rstd, gamma, beta, cx);
  float acc[1][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = 0.f;
  const int p0 = blockIdx.x * g.PY * ppt + py;
  for (int k = 0; k < ppt; ++k) {
    const int p = p0 + k * g.PY;
    if (p >= g.hw) break;
    const int o = pix_off(g, b, p, cx);
    float f[8], u[8];
    unpack8(*(const u32x4_t*)(x + o), f);
    if (MUL) unpack8(*(const u32x4_t*)(gu + o), u);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = act_fwd<ACT>(f[i] * bn.sc[i] + bn.sh[i]);
      if (MUL) v *= u[i];
      acc[0][i] += v;
    }
  }
  float* dst = out + (size_t)blockIdx.x * row_stride + (size_t)b * g.C;
  block_fold<1>(acc, cx, py, g.c8, g.PY, lds, [&](int, int c, float s) { atomicAdd(dst + c, s * scale); });
}

// gradient entering the activation:  g_a = SE ? gu*gate[b,c] + gpool[b,c]/HW : (POOL ? gpooled[b,c]/HW : gu)
// backward pass 1: per-channel sums of g_y = g_a * act'(y) and g_y * xhat  -> 32-slot scratch
template <int ACT, bool SE, bool POOL>
__global__ __launch_bounds__(kThreads) void mb_bwd_reduce_kernel(
    const bf16_t* __restrict__ gu, const float* __restrict__ gate, const float* __restrict__ gpool,
    const bf16_t* __restrict__ x, const float* mean, const float* rstd, const float* gamma, const float* beta,
    MbGeom g, int ppt, float* __restrict__ scratch, unsigned slot_mask) {
  extern __shared__ float lds[];
  const int cx = threadIdx.x % g.c8, py = threadIdx.x / g.c8, b = blockIdx.y;
  BnRegs bn;
  load_bn(bn, mean, rstd, gamma, beta, cx);
  float mu[8], rs[8], gt[8], gp[8];
  const float inv_hw = 1.f / (float)g.hw;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cx * 8 + i;
    mu[i] = mean[c];
    rs[i] = rstd[c];
    if (SE) gt[i] = gate[(size_t)b * g.C + c];
    if (SE || POOL) gp[i] = gpool[(size_t)b * g.C + c] * inv_hw;
  }
  float acc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.f;
  const int p0 = blockIdx.x * g.PY * ppt + py;
  for (int k = 0; k < ppt; ++k) {
    const int p = p0 + k * g.PY;
    if (p >= g.hw) break;
    const int o = pix_off(g, b, p, cx);
    float fx[8], fg[8];
    unpack8(*(const u32x4_t*)(x + o), fx);
    if (!POOL) unpack8(*(const u32x4_t*)(gu + o), fg);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float ga = POOL ? gp[i] : fg[i];
      if (SE) ga = ga * gt[i] + gp[i];
      const float gy = ga * act_bwd<ACT>(fx[i] * bn.sc[i] + bn.sh[i]);
      acc[0][i] += gy;
      acc[1][i] += gy * ((fx[i] - mu[i]) * rs[i]);
    }
  }
  const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) & slot_mask;   // (deterministic mode: a row per block)
  block_fold<2>(acc, cx, py, g.c8, g.PY, lds, [&](int q, int c, float s) {
    atomicAdd(scratch + ((size_t)slot * 2 + q) * g.C + c, s);
  });
}

// backward pass 2: gx = gamma*rstd * (g_y - mean(g_y) - xhat * mean(g_y*xhat)) [+ gx_add]
template <int ACT, bool SE, bool POOL, bool HAS_ADD>
__global__ __launch_bounds__(kThreads) void mb_bwd_apply_kernel(
    const bf16_t* __restrict__ gu, const float* __restrict__ gate, const float* __restrict__ gpool,
    const bf16_t* __restrict__ x, const float* mean, const float* rstd, const float* gamma, const float* beta,
    const float* __restrict__ dsum, const bf16_t* __restrict__ gx_add, MbGeom g, int ppt,
    bf16_t* __restrict__ gx) {
  const int cx = threadIdx.x % g.c8, py = threadIdx.x / g.c8, b = blockIdx.y;
  BnRegs bn;
  load_bn(bn, mean, rstd, gamma, beta, cx);
  float mu[8], rs[8], gt[8], gp[8], k0[8], k1[8];
  const float inv_hw = 1.f / (float)g.hw;
  const float inv_n = 1.f / ((float)g.hw * (float)g.B);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cx * 8 + i;
    mu[i] = mean[c];
    rs[i] = rstd[c];
    k0[i] = dsum[c] * inv_n;
    k1[i] = dsum[g.C + c] * inv_n;
    if (SE) gt[i] = gate[(size_t)b * g.C + c];
    if (SE || POOL) gp[i] = gpool[(size_t)b * g.C + c] * inv_hw;
  }
  const int p0 = blockIdx.x * g.PY * ppt + py;
  for (int k = 0; k < ppt; ++k) {
    const int p = p0 + k * g.PY;
    if (p >= g.hw) break;
    const int o = pix_off(g, b, p, cx);
    float fx[8], fg[8], fa[8], out[8];
    unpack8(*(const u32x4_t*)(x + o), fx);
    if (!POOL) unpack8(*(const u32x4_t*)(gu + o), fg);
    if (HAS_ADD) unpack8(*(const u32x4_t*)(gx_add + o), fa);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float ga = POOL ? gp[i] : fg[i];
      if (SE) ga = ga * gt[i] + gp[i];
      const float gy = ga * act_bwd<ACT>(fx[i] * bn.sc[i] + bn.sh[i]);
      const float xh = (fx[i] - mu[i]) * rs[i];
      float v = bn.sc[i] * (gy - k0[i] - xh * k1[i]);
      if (HAS_ADD) v += fa[i];
      out[i] = v;
    }
    *(u32x4_t*)(gx + o) = pack8(out);
  }
}

__global__ __launch_bounds__(256) void mb_bwd_finalize_kernel(float* __restrict__ scratch, int C,
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
      scratch[((size_t)k * 2 + 0) * C + c] = 0.f;   // zero-on-entry contract for the next user
      scratch[((size_t)k * 2 + 1) * C + c] = 0.f;
    }
    dsum[c] = s0;
    dsum[C + c] = s1;
    if (dbeta) dbeta[c] += s0;
    if (dgamma) dgamma[c] += s1;
  }
}

// Squeeze-and-excitation backward in ONE pass over (gu, x).  The gradient entering the activation of the depthwise
// output is g_a = gu * gate[b,c] + gpool[b,c] / HW, and gpool comes out of the SE branch's backward, which needs
// dL/dgate[b,c] = sum_hw gu * act(y) first -- so rounds 1-5 read (gu, x) three times: dL/dgate, then the BatchNorm
// sums of g_y = g_a * act'(y), then the elementwise pass.  But g_y is LINEAR in (gate, gpool) per image and channel:
//   sum g_y      = sum_b gate[b,c] * S1[b,c] + gpool[b,c] / HW * S3[b,c]        S1 = sum_hw gu * act'(y)   S3 = sum_hw act'(y)
//   sum g_y*xhat = sum_b gate[b,c] * S2[b,c] + gpool[b,c] / HW * S4[b,c]        S2 = sum_hw gu * act'(y) * xhat   S4 = sum_hw act'(y) * xhat
// so this kernel adds S0 = dL/dgate and S1..S4 per (image, channel) into sums[5][B][C] (zero on entry) and
// mb_se_finalize_kernel turns them into the BatchNorm sums once gpool exists -- and zeroes the buffer again for the next
// user, like the BatchNorm slots' finalize kernels do: one pass over the two tensors less, no memset launch.
template <int ACT>
__global__ __launch_bounds__(kThreads) void mb_se_sums_kernel(const bf16_t* __restrict__ gu,
                                                              const bf16_t* __restrict__ x, const float* mean,
                                                              const float* rstd, const float* gamma,
                                                              const float* beta, MbGeom g, int ppt,
                                                              float* __restrict__ sums) {
  extern __shared__ float lds[];
  const int cx = threadIdx.x % g.c8, py = threadIdx.x / g.c8, b = blockIdx.y;
  BnRegs bn;
  load_bn(bn, mean, rstd, gamma, beta, cx);
  float mu[8], rs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    mu[i] = mean[cx * 8 + i];
    rs[i] = rstd[cx * 8 + i];
  }
  float acc[5][8];
#pragma unroll
  for (int q = 0; q < 5; ++q)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[q][i] = 0.f;
  const int p0 = blockIdx.x * g.PY * ppt + py;
  for (int k = 0; k < ppt; ++k) {
    const int p = p0 + k * g.PY;
    if (p >= g.hw) break;
    const int o = pix_off(g, b, p, cx);
    float fx[8], fg[8];
    unpack8(*(const u32x4_t*)(x + o), fx);
    unpack8(*(const u32x4_t*)(gu + o), fg);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float y = fx[i] * bn.sc[i] + bn.sh[i];
      const float xh = (fx[i] - mu[i]) * rs[i];
      const float d = act_bwd<ACT>(y), gd = fg[i] * d;
      acc[0][i] += fg[i] * act_fwd<ACT>(y);
      acc[1][i] += gd;
      acc[2][i] += gd * xh;
      acc[3][i] += d;
      acc[4][i] += d * xh;
    }
  }
  const size_t plane = (size_t)g.B * g.C;
  float* dst = sums + (size_t)b * g.C;
  block_fold<5>(acc, cx, py, g.c8, g.PY, lds, [&](int q, int c, float s) { atomicAdd(dst + q * plane + c, s); });
}

// sums[5][B][C] + gate, gpool -> the BatchNorm-backward sums (dsum[0..C) = sum g_y, dsum[C..2C) = sum g_y * xhat; dbeta,
// dgamma accumulate).  64 channels x 4 image lanes per block; a channel's sum is its 4 lane partials in lane order.
__global__ __launch_bounds__(256) void mb_se_finalize_kernel(float* __restrict__ sums,
                                                             const float* __restrict__ gate,
                                                             const float* __restrict__ gpool, int B, int C,
                                                             float inv_hw, float* __restrict__ dsum,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float part[2][4][64];
  const int cl = threadIdx.x & 63, lane = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  const size_t plane = (size_t)B * C;
  float s0 = 0.f, s1 = 0.f;
  if (c < C)
#pragma unroll 4
    for (int b = lane; b < B; b += 4) {
      const size_t o = (size_t)b * C + c;
      const float gt = gate[o], gp = gpool[o] * inv_hw;
      s0 += gt * sums[plane + o] + gp * sums[3 * plane + o];
      s1 += gt * sums[2 * plane + o] + gp * sums[4 * plane + o];
#pragma unroll
      for (int q = 0; q < 5; ++q) sums[q * plane + o] = 0.f;      // (dL/dgate, plane 0, was consumed by nbdt_se_gate_bwd)
    }
  part[0][lane][cl] = s0;
  part[1][lane][cl] = s1;
  __syncthreads();
  if (lane == 0 && c < C) {
    const float t0 = ((part[0][0][cl] + part[0][1][cl]) + part[0][2][cl]) + part[0][3][cl];
    const float t1 = ((part[1][0][cl] + part[1][1][cl]) + part[1][2][cl]) + part[1][3][cl];
    dsum[c] = t0;
    dsum[C + c] = t1;
    if (dbeta) dbeta[c] += t0;
    if (dgamma) dgamma[c] += t1;
  }
}

// ------------------------------------------------------------------------------------------
// depthwise convolution, weights fp32 [k*k][C]; x padded [B][H+2][W+2][C], y padded [B][Ho+2][Wo+2][C]
struct DwGeom {
  MbGeom out;            // geometry of the OUTPUT tensor (Ho, Wo)
  int Hi, Wi, rowi, imgi;
  int k, stride, pad;
};


// Row-segment form of the depthwise convolution (forward for stride 1|2, and -- with FLIP -- the
// stride-1 data gradient, which is the same correlation with the kernel rotated by 180 degrees):
// a thread owns 8 channels x TW consecutive outputs of one row; per kernel row it loads the
// TW*S + K - S input pixels under them ONCE and feeds each to every output it overlaps
// (K=5: 7.5 loads per output instead of 25), K x 8 weights of the row in registers.
// STATS: 0 none; 1 = sum / sum of squares of the (rounded) outputs, the batch statistics of the NEXT BatchNorm;
// 2 (data gradient only) = the backward sums of the BatchNorm + swish that PRODUCED this conv's input: the output is
// dL/d(swish(bn(bx))), so with g' = out * swish'(bn(bx)) the kernel accumulates sum(g'), sum(g' * xhat) while it still
// holds the tile -- the separate reduction pass (a re-read of this output and of bx) disappears, bx is read once here.
struct DwBn {
  const bf16_t* x;     // the BatchNorm's input (same geometry as this launch's output)
  const float *mean, *rstd, *gamma, *beta;
};
template <int K, int S, int TW, bool FLIP, int STATS>
__global__ __launch_bounds__(kThreads) void dw_row_kernel(const bf16_t* __restrict__ x,
                                                          const float* __restrict__ w, DwGeom d, int nseg,
                                                          int PY, bf16_t* __restrict__ y,
                                                          float* __restrict__ stats, unsigned slot_mask, DwBn bnb) {
  extern __shared__ float lds[];
  const MbGeom& g = d.out;
  constexpr int PAD = K / 2, SPAN = TW * S + K - S;
  const unsigned lb = xcd_contiguous(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);   // (image, row block)
  const int cx = threadIdx.x % g.c8, py = threadIdx.x / g.c8, b = (int)(lb / gridDim.x);
  int item = (int)(lb % gridDim.x) * PY + py;
  const bool live = item < g.H * nseg;
  if (!STATS && !live) return;
  item = live ? item : 0;                    // STATS: idle threads still take part in the block fold
  const int ho = item / nseg;
  const int wo0 = (item - ho * nseg) * TW;
  float acc[TW][8];
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const int hi = ho * S + r - PAD;
    if (hi < -1 || hi > d.Hi) continue;                  // outside even the zero border
    float wr[K][8];
#pragma unroll
    for (int sx = 0; sx < K; ++sx) {
      const int tap = FLIP ? (K - 1 - r) * K + (K - 1 - sx) : r * K + sx;
      const float4 w0 = *(const float4*)(w + (size_t)tap * g.C + cx * 8);
      const float4 w1 = *(const float4*)(w + (size_t)tap * g.C + cx * 8 + 4);
      wr[sx][0] = w0.x; wr[sx][1] = w0.y; wr[sx][2] = w0.z; wr[sx][3] = w0.w;
      wr[sx][4] = w1.x; wr[sx][5] = w1.y; wr[sx][6] = w1.z; wr[sx][7] = w1.w;
    }
    const bf16_t* xrow = x + (size_t)b * d.imgi + (hi + 1) * d.rowi + g.C + cx * 8;   // + wi * C
#pragma unroll
    for (int j = 0; j < SPAN; ++j) {
      const int wi = wo0 * S - PAD + j;
      if (wi < -1 || wi > d.Wi) continue;
      float f[8];
      unpack8(*(const u32x4_t*)(xrow + wi * g.C), f);
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        const int sx = j - t * S;                          // static after unrolling
        if (sx >= 0 && sx < K) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[t][i] += f[i] * wr[sx][i];
        }
      }
    }
  }
  const size_t row_off = (size_t)b * g.img + (ho + 1) * g.row + g.C + cx * 8;
  bf16_t* yrow = y + row_off;
  float st[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) st[0][i] = st[1][i] = 0.f;
  float bsc[8], bsh[8], bmu[8], brs[8];
  if (STATS == 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cx * 8 + i;
      bmu[i] = bnb.mean[c];
      brs[i] = bnb.rstd[c];
      bsc[i] = bnb.gamma[c] * brs[i];
      bsh[i] = bnb.beta[c] - bmu[i] * bsc[i];
    }
  }
#pragma unroll
  for (int t = 0; t < TW; ++t)
    if (live && wo0 + t < g.W) {
      const u32x4_t v = pack8(acc[t]);
      *(u32x4_t*)(yrow + (wo0 + t) * g.C) = v;
      if (STATS == 1) {      // batch statistics of the NEXT BatchNorm, from the rounded values it will read
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { st[0][i] += f[i]; st[1][i] += f[i] * f[i]; }
      }
      if (STATS == 2) {      // backward sums of the producing BatchNorm + swish, from the rounded gradient
        float f[8], fx[8];
        unpack8(v, f);
        unpack8(*(const u32x4_t*)(bnb.x + row_off + (wo0 + t) * g.C), fx);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float gy = f[i] * act_bwd<NBDT_ACT_SWISH>(fx[i] * bsc[i] + bsh[i]);
          st[0][i] += gy;
          st[1][i] += gy * ((fx[i] - bmu[i]) * brs[i]);
        }
      }
    }
  if (STATS) {
    const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) & slot_mask;   // (deterministic mode: a row per block)
    block_fold<2>(st, cx, py, g.c8, PY, lds, [&](int q, int c, float v) {
      atomicAdd(stats + ((size_t)slot * 2 + q) * g.C + c, v);
    });
  }
}

// Deterministic mode (nbdt_set_deterministic): where a reduction kernel of `blocks` blocks adds its per-block sums of n
// floats -- the caller's 32-slot scratch, or a zeroed library-owned row per block that slot_finish() sums in block
// order into slot 0 of the caller's scratch (same scheme as csrc/bn.hip).
struct SlotTarget {
  float* ptr;
  unsigned mask;
  bool det;
};
static int slot_target(hipStream_t st, float* scratch, int blocks, size_t n, SlotTarget* t) {
  t->ptr = scratch; t->mask = kSlots - 1; t->det = false;
  if (!deterministic()) return NBDT_OK;
  float* rows = det_rows(st, (size_t)blocks * n);
  if (!rows) return nbdt::fail(NBDT_ENOMEM, "deterministic mode: %s (%s)", "no workspace for the per-block rows", nbdt::det_rows_why());
  NBDT_HIP_CHECK(hipMemsetAsync(rows, 0, (size_t)blocks * n * sizeof(float), st));
  t->ptr = rows; t->mask = ~0u; t->det = true;
  return NBDT_OK;
}
static int slot_finish(hipStream_t st, const SlotTarget& t, int blocks, size_t n, float* scratch) {
  return t.det ? det_fold(st, t.ptr, blocks, n, scratch) : NBDT_OK;
}

template <int K, int S, bool FLIP>
static int launch_dw_row(const void* x, const float* w, const DwGeom& d, int B, void* y, float* stats_out,
                         hipStream_t st, const DwBn* bn = nullptr) {
  const int Wo = d.out.W, Ho = d.out.H;
  const int tw = (Wo % 7 == 0) ? 7 : (Wo >= 8 ? 8 : 4);
  const int nseg = (Wo + tw - 1) / tw;
  int PY = kThreads / d.out.c8;
  if (PY < 1) PY = 1;
  if (PY > Ho * nseg) PY = Ho * nseg;
  const dim3 grid((Ho * nseg + PY - 1) / PY, B), blk(d.out.c8 * PY);
  SlotTarget tgt{stats_out, kSlots - 1, false};
  const int nblk = (int)(grid.x * grid.y);
  if (stats_out) {
    const int rc = slot_target(st, stats_out, nblk, 2 * (size_t)d.out.C, &tgt);
    if (rc) return rc;
  }
  float* const stats = tgt.ptr;
  const unsigned slot_mask = tgt.mask;
  const size_t shmem = stats ? (size_t)d.out.c8 * PY * 16 * sizeof(float) : 0;
#define NBDT_GO(TW)                                                                                              \
  do {                                                                                                           \
    if (stats && bn) hipLaunchKernelGGL((dw_row_kernel<K, S, TW, FLIP, 2>), grid, blk, shmem, st, (const bf16_t*)x, w, d, nseg, PY, (bf16_t*)y, stats, slot_mask, *bn); \
    else if (stats) hipLaunchKernelGGL((dw_row_kernel<K, S, TW, FLIP, 1>), grid, blk, shmem, st, (const bf16_t*)x, w, d, nseg, PY, (bf16_t*)y, stats, slot_mask, DwBn{}); \
    else hipLaunchKernelGGL((dw_row_kernel<K, S, TW, FLIP, 0>), grid, blk, 0, st, (const bf16_t*)x, w, d, nseg, PY, (bf16_t*)y, stats, slot_mask, DwBn{}); \
  } while (0)
  if (tw == 7) NBDT_GO(7); else if (tw == 8) NBDT_GO(8); else NBDT_GO(4);
#undef NBDT_GO
  NBDT_LAUNCH_CHECK();
  return stats_out ? slot_finish(st, tgt, nblk, 2 * (size_t)d.out.C, stats_out) : NBDT_OK;
}

// gx[hi][wi] = sum_{r,s : (hi+pad-r) % stride == 0, ...} gy[(hi+pad-r)/stride][(wi+pad-s)/stride] * w[r][s]
// `in` = geometry of the INPUT-sized gradient being written
__global__ __launch_bounds__(kThreads) void dw_bwd_data_kernel(const bf16_t* __restrict__ gy,
                                                               const float* __restrict__ w, MbGeom in, int Ho,
                                                               int Wo, int k, int stride, int pad, int ppt,
                                                               bf16_t* __restrict__ gx) {
  const unsigned lb = xcd_contiguous(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
  const int cx = threadIdx.x % in.c8, py = threadIdx.x / in.c8, b = (int)(lb / gridDim.x);
  const int rowo = (Wo + 2) * in.C, imgo = (Ho + 2) * rowo;
  const int p0 = (int)(lb % gridDim.x) * in.PY * ppt + py;
  for (int q = 0; q < ppt; ++q) {
    const int p = p0 + q * in.PY;
    if (p >= in.hw) break;
    const int hi = (int)fdiv((unsigned)p, in.div_w);
    const int wi = p - hi * in.W;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int r = 0; r < k; ++r) {
      const int th = hi + pad - r;
      if (th < 0 || (stride == 2 && (th & 1))) continue;
      const int ho = stride == 2 ? th >> 1 : th;
      if (ho >= Ho) continue;
      for (int s = 0; s < k; ++s) {
        const int tw = wi + pad - s;
        if (tw < 0 || (stride == 2 && (tw & 1))) continue;
        const int wo = stride == 2 ? tw >> 1 : tw;
        if (wo >= Wo) continue;
        float f[8];
        unpack8(*(const u32x4_t*)(gy + (size_t)b * imgo + (ho + 1) * rowo + (wo + 1) * in.C + cx * 8), f);
        const float4 w0 = *(const float4*)(w + (size_t)(r * k + s) * in.C + cx * 8);
        const float4 w1 = *(const float4*)(w + (size_t)(r * k + s) * in.C + cx * 8 + 4);
        acc[0] += f[0] * w0.x; acc[1] += f[1] * w0.y; acc[2] += f[2] * w0.z; acc[3] += f[3] * w0.w;
        acc[4] += f[4] * w1.x; acc[5] += f[5] * w1.y; acc[6] += f[6] * w1.z; acc[7] += f[7] * w1.w;
      }
    }
    *(u32x4_t*)(gx + pix_off(in, b, p, cx)) = pack8(acc);
  }
}

// Row-segment form of the STRIDE-2 data gradient (round 5).  dw_bwd_data_kernel above gathers, per input pixel, the
// (K/2 .. K/2+1)^2 gradient pixels that reach it -- 6.25 16-byte loads per output at K = 5, each its own round trip -- and
// took 2.2x the time of its bytes (97 us per launch on EfficientNet-B0's four stride-2 layers).  Here a thread owns 8
// channels x TW consecutive pixels of one input row (TW even, so the segment starts on an even column): for each kernel row
// r of the right parity it loads the TW/2 + K/2 + 1 gradient pixels of output row (hi + pad - r)/2 that reach the segment
// ONCE and feeds each to the outputs it reaches (tap s = t + pad - 2 (wo - wi0/2), static after unrolling).  The gradient
// tensor's zero border stands in for wo = -1 and wo = Wo.  Same sums in a different order than the gather kernel.
template <int K, int TW>
__global__ __launch_bounds__(kThreads) void dw_bwd_data_s2_row_kernel(const bf16_t* __restrict__ gy,
                                                                      const float* __restrict__ w, MbGeom in, int Ho,
                                                                      int Wo, int nseg, int PY, bf16_t* __restrict__ gx) {
  // gradient columns wo = wi0/2 - 1 + j that reach the segment: tap s = t + PAD + 2 - 2 j in [0, K) for some t in [0, TW)
  constexpr int PAD = K / 2, J0 = PAD == 1 ? 1 : 0, J1 = (TW + PAD + 1) / 2, SPAN = J1 - J0 + 1;
  const unsigned lb = xcd_contiguous(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
  const int cx = threadIdx.x % in.c8, py = threadIdx.x / in.c8, b = (int)(lb / gridDim.x);
  const int item = (int)(lb % gridDim.x) * PY + py;
  if (item >= in.H * nseg) return;
  const int hi = item / nseg;
  const int wi0 = (item - hi * nseg) * TW;
  const int rowo = (Wo + 2) * in.C, imgo = (Ho + 2) * rowo;
  float acc[TW][8];
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const int th = hi + PAD - r;
    if (th < 0 || (th & 1) || (th >> 1) >= Ho) continue;
    const int ho = th >> 1;
    float wr[K][8];
#pragma unroll
    for (int sx = 0; sx < K; ++sx) {
      const float4 w0 = *(const float4*)(w + (size_t)(r * K + sx) * in.C + cx * 8);
      const float4 w1 = *(const float4*)(w + (size_t)(r * K + sx) * in.C + cx * 8 + 4);
      wr[sx][0] = w0.x; wr[sx][1] = w0.y; wr[sx][2] = w0.z; wr[sx][3] = w0.w;
      wr[sx][4] = w1.x; wr[sx][5] = w1.y; wr[sx][6] = w1.z; wr[sx][7] = w1.w;
    }
    const bf16_t* grow = gy + (size_t)b * imgo + (ho + 1) * rowo + in.C + cx * 8;     // + wo * C; wo = -1 and Wo: zero border
    u32x4_t raw[SPAN];
#pragma unroll
    for (int j = 0; j < SPAN; ++j) {
      const int wo = (wi0 >> 1) - 1 + J0 + j;
      const u32x4_t zero4 = {0u, 0u, 0u, 0u};
      raw[j] = (wo >= -1 && wo <= Wo) ? *(const u32x4_t*)(grow + wo * in.C) : zero4;
    }
#pragma unroll
    for (int j = 0; j < SPAN; ++j) {
      float f[8];
      unpack8(raw[j], f);
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        const int sx = t + PAD + 2 - 2 * (J0 + j);         // wi0 + t + pad - 2 wo, static after unrolling
        if (sx >= 0 && sx < K) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[t][i] += f[i] * wr[sx][i];
        }
      }
    }
  }
  bf16_t* xrow = gx + (size_t)b * in.img + (hi + 1) * in.row + in.C + cx * 8;
#pragma unroll
  for (int t = 0; t < TW; ++t)
    if (wi0 + t < in.W) *(u32x4_t*)(xrow + (wi0 + t) * in.C) = pack8(acc[t]);
}

// dw[r*K+s][c] += sum_{b, output pixels} gy * x[ho*S + r - pad][wo*S + s - pad]
// thread = (8-channel chunk, output row ho), blockIdx.z = kernel row r.  The thread walks its row left to
// right keeping the K input pixels under the current output in registers (a sliding window: S new
// 16-byte loads per output pixel instead of K), K x 8 fp32 accumulators.
// A block is CB channel chunks (at most 32: one 512-byte run per pixel) x PY output rows; blockIdx.x = (row block,
// channel block).  Wide layers used to put ALL their channel chunks into a block -- 1152 channels: 144 threads, ONE
// row -- so a block folded one row of 18 images before its K*8 atomics per thread, and the chip held 630 waves, each a
// serial walk over 126 pixels.  With the channels split, 7 rows fold in the block and the batch chunk shrinks instead.
template <int K, int S, int U>
__global__ __launch_bounds__(kThreads) void dw_bwd_weight_kernel(const bf16_t* __restrict__ x,
                                                                 const bf16_t* __restrict__ gy, DwGeom d, int PY,
                                                                 int CB, int cblocks, int bchunk,
                                                                 float* __restrict__ dw, long long row_stride) {
  extern __shared__ float lds[];
  const MbGeom& g = d.out;
  // logical order: kernel row r fastest (the K blocks of one (rows, channels, batch chunk) read the same gradient rows and
  // input rows one apart), then the x index, then the batch chunk -- contiguous per XCD (xcd_contiguous)
  const unsigned lb = xcd_contiguous(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                     gridDim.x * gridDim.y * gridDim.z);
  const int r = (int)(lb % K), bx = (int)((lb / K) % gridDim.x), by = (int)(lb / (K * gridDim.x));
  const int cb = bx % cblocks, rb = bx / cblocks;
  const int cxl = threadIdx.x % CB, py = threadIdx.x / CB;
  const int cx = cb * CB + cxl;
  const bool live = cx < g.c8;
  constexpr int PAD = K / 2;
  float acc[K][8];
#pragma unroll
  for (int s = 0; s < K; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[s][i] = 0.f;
  const int ho = rb * PY + py;
  const int hi = ho * S + r - PAD;
  const int b0 = by * bchunk;
  const int b1 = b0 + bchunk < g.B ? b0 + bchunk : g.B;
  if (live && ho < g.H && hi >= 0 && hi < d.Hi) {
    for (int b = b0; b < b1; ++b) {    // images of this block's batch chunk: one fold + atomics for all
      const bf16_t* xrow = x + (size_t)b * d.imgi + (hi + 1) * d.rowi + g.C + cx * 8;   // + wi * C
      const bf16_t* grow = gy + (size_t)b * g.img + (ho + 1) * g.row + g.C + cx * 8;    // + wo * C
      float win[K][8];
      const u32x4_t zero4 = {0u, 0u, 0u, 0u};
      auto raw = [&](int wi) -> u32x4_t {      // inside the padded row (the border itself is zero), else zero
        return (wi >= -1 && wi <= d.Wi) ? *(const u32x4_t*)(xrow + wi * g.C) : zero4;
      };
#pragma unroll
      for (int j = 0; j < K; ++j) unpack8(raw(j - PAD), win[j]);
      // U output pixels per trip: their gradient vectors and the S new window columns each of them brings are all
      // requested before the first is used (U * (1 + S) 16-byte loads in flight per thread; one at a time, the walk
      // was a chain of load round trips -- 124 us for the 2 x 48 MB of a 28x28x240 layer, profiles/r04_dw_wgrad.txt)
      for (int wo0 = 0; wo0 < g.W; wo0 += U) {
        u32x4_t rg[U], rx[U][S];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int wo = wo0 + u;
          rg[u] = wo < g.W ? *(const u32x4_t*)(grow + wo * g.C) : zero4;
#pragma unroll
          for (int j = 0; j < S; ++j) rx[u][j] = wo + 1 < g.W ? raw((wo + 1) * S - PAD + (K - S) + j) : zero4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (wo0 + u < g.W) {
            float fg[8];
            unpack8(rg[u], fg);
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[j][i] += fg[i] * win[j][i];
#pragma unroll
            for (int j = 0; j + S < K; ++j)
#pragma unroll
              for (int i = 0; i < 8; ++i) win[j][i] = win[j + S][i];
#pragma unroll
            for (int j = 0; j < S; ++j) unpack8(rx[u][j], win[K - S + j]);
          }
        }
      }
    }
  }
  // row_stride: 0, or (deterministic mode) K*K*C: a zeroed copy of dw per (row block, batch chunk), folded in order
  float* dst = dw + (size_t)(bx + by * gridDim.x) * row_stride + (size_t)r * K * g.C;
  block_fold<K>(acc, cxl, py, CB, PY, lds, [&](int s, int c, float v) {
    if (cb * CB * 8 + c < g.C) atomicAdd(dst + (size_t)s * g.C + cb * CB * 8 + c, v);   // (the last channel block may be short)
  });
}

// ------------------------------------------------------------------------------------------
// squeeze-and-excitation gate.  One block per sample.
//   hidden = swish(W1 pooled + b1)  [S];   gate = sigmoid(W2 hidden + b2)  [Cr];  gate[c >= Cr] = 0
// W1 [S][Cr], W2 [Cr][S] (the 1x1 convs with bias of pytorchcv SEBlock), pooled/gate rows have stride C.
__global__ __launch_bounds__(1024) void se_gate_fwd_kernel(const float* __restrict__ pooled,
                                                          const float* __restrict__ w1,
                                                          const float* __restrict__ b1,
                                                          const float* __restrict__ w2,
                                                          const float* __restrict__ b2, int C, int Cr, int S,
                                                          float* __restrict__ pre1, float* __restrict__ gate) {
  extern __shared__ float lds[];   // pooled[Cr] | h[S]
  float* pl = lds;
  float* hl = lds + Cr;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nthr = blockDim.x, nwave = blockDim.x >> 6;
  for (int c = threadIdx.x; c < Cr; c += nthr) pl[c] = pooled[(size_t)b * C + c];
  __syncthreads();
  for (int s = wave; s < S; s += nwave) {
    float acc = 0.f;
#pragma unroll 8
    for (int c = lane; c < Cr; c += 64) acc += w1[(size_t)s * Cr + c] * pl[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
      const float v = acc + b1[s];
      pre1[(size_t)b * S + s] = v;
      hl[s] = v * sigmoidf_(v);
    }
  }
  __syncthreads();
  // (a thread per row of W2 [Cr][S] walking s: the row is 4 S contiguous bytes that stay in L1 across the walk -- a wave
  // per row with lanes along s measured 2-4x slower, profiles/r04_dw_wgrad.txt)
  for (int c = threadIdx.x; c < C; c += nthr) {
    float v = 0.f;
    if (c < Cr) {
      float acc = b2[c];
#pragma unroll 8
      for (int s = 0; s < S; ++s) acc += w2[(size_t)c * S + s] * hl[s];
      v = sigmoidf_(acc);
    }
    gate[(size_t)b * C + c] = v;
  }
}

// per sample: dgate[b,c] (= sum_hw gu*a) -> dpre2[b,c], dpre1[b,s], gpool[b,c] (gradient of the pooled mean)
__global__ __launch_bounds__(1024) void se_gate_bwd_kernel(const float* __restrict__ dgate,
                                                          const float* __restrict__ gate,
                                                          const float* __restrict__ pre1,
                                                          const float* __restrict__ w1,
                                                          const float* __restrict__ w2, int C, int Cr, int S,
                                                          float* __restrict__ dpre2, float* __restrict__ dpre1,
                                                          float* __restrict__ gpool) {
  extern __shared__ float lds[];   // d2[Cr] | d1[S] | per-wave partials [nwave][S]
  float* d2 = lds;
  float* d1 = lds + Cr;
  float* part = d1 + S;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nthr = blockDim.x, nwave = blockDim.x >> 6;
  for (int c = threadIdx.x; c < Cr; c += nthr) {
    const float gt = gate[(size_t)b * C + c];
    const float v = dgate[(size_t)b * C + c] * gt * (1.f - gt);
    d2[c] = v;
    dpre2[(size_t)b * Cr + c] = v;
  }
  __syncthreads();
  // sum_c W2[c][s] d2[c].  Wide layers (16 waves): a wave walks rows c = wave, wave + 16, ... with its lanes along s
  // (contiguous reads of a row) and the waves' partial vectors are added in wave order -- 60 -> 44 us at 1152 x 48, where
  // lanes along c read W2 with stride S, 64 lines per load, 18 times per s.  Narrow layers keep the wave-per-s form (it
  // measures the same or better there).
  if (nwave > 4) {
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s2 = s0 + lane;
      float acc = 0.f;
      if (s2 < S) {
#pragma unroll 8
        for (int c = wave; c < Cr; c += nwave) acc += w2[(size_t)c * S + s2] * d2[c];
      }
      if (s2 < S) part[wave * S + s2] = acc;
    }
    __syncthreads();
  }
  for (int s = nwave > 4 ? threadIdx.x : wave; s < S; s += nwave > 4 ? nthr : nwave) {
    float acc = 0.f;
    if (nwave > 4) {
      for (int wv = 0; wv < nwave; ++wv) acc += part[wv * S + s];
    } else {
#pragma unroll 8
      for (int c = lane; c < Cr; c += 64) acc += w2[(size_t)c * S + s] * d2[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (lane != 0) continue;
    }
    const float p = pre1[(size_t)b * S + s];
    const float sg = sigmoidf_(p);
    const float v = acc * sg * (1.f + p * (1.f - sg));
    d1[s] = v;
    dpre1[(size_t)b * S + s] = v;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += nthr) {
    float acc = 0.f;
    if (c < Cr) {
#pragma unroll 8
      for (int s = 0; s < S; ++s) acc += w1[(size_t)s * Cr + c] * d1[s];
    }
    gpool[(size_t)b * C + c] = acc;
  }
}

// parameter gradients of the two SE projections, summed over the batch: thread per (c, s) pair,
// blockIdx.y = batch chunk (independent loads unrolled; one atomic per output per chunk)
__global__ __launch_bounds__(256) void se_param_grad_kernel(const float* __restrict__ dpre2,
                                                            const float* __restrict__ dpre1,
                                                            const float* __restrict__ pre1,
                                                            const float* __restrict__ pooled, int B, int C, int Cr,
                                                            int S, int bchunk, float* __restrict__ dw1,
                                                            float* __restrict__ db1, float* __restrict__ dw2,
                                                            float* __restrict__ db2) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Cr * S) return;
  // the thread owns TWO outputs, each where its neighbours' are contiguous: dw1[s][c] with consecutive threads along c,
  // dw2[c2][s2] (= dw2 + idx) with consecutive threads along s -- one (c, s) pair for both put the dw2 atomics S floats
  // apart, 64 sectors per instruction
  const int c = idx % Cr, s = idx / Cr;
  const int s2 = idx % S, c2 = idx / S;
  const int b0 = blockIdx.y * bchunk;
  const int b1 = b0 + bchunk < B ? b0 + bchunk : B;
  float a1 = 0.f, a2 = 0.f, sb1 = 0.f, sb2 = 0.f;
#pragma unroll 4
  for (int b = b0; b < b1; ++b) {
    const float p = pre1[(size_t)b * S + s2];
    const float e1 = dpre1[(size_t)b * S + s], e2 = dpre2[(size_t)b * Cr + c2];
    const float pl = pooled[(size_t)b * C + c];
    const float h = p * sigmoidf_(p);
    a1 += e1 * pl;
    a2 += e2 * h;
    sb1 += e1;
    sb2 += e2;
  }
  atomicAdd(dw1 + idx, a1);                     // = dw1 + s * Cr + c
  atomicAdd(dw2 + idx, a2);                     // = dw2 + c2 * S + s2
  if (c == 0) atomicAdd(db1 + s, sb1);
  if (s2 == 0) atomicAdd(db2 + c2, sb2);
}

// ------------------------------------------------------------------------------------------
// dropout on [n] fp32: keep-mask from a counter hash (seed, element index); y = x * mask / (1-p)
__device__ __forceinline__ unsigned hash32(unsigned v) {
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, long long n, float p,
                                                      float inv_keep, unsigned seed,
                                                      unsigned char* __restrict__ mask, float* __restrict__ y) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned h = hash32((unsigned)i * 0x9e3779b9u + hash32(seed));
  const bool keep = (float)(h >> 8) * (1.f / 16777216.f) >= p;
  mask[i] = keep ? 1 : 0;
  y[i] = keep ? x[i] * inv_keep : 0.f;
}
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ gy, long long n, float inv_keep,
                                                          const unsigned char* __restrict__ mask,
                                                          float* __restrict__ gx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  gx[i] = mask[i] ? gy[i] * inv_keep : 0.f;
}

constexpr int kPpt = 8;

}  // namespace

// ------------------------------------------------------------------------------------------ host

#define NBDT_ACT_SWITCH(act, MACRO)                                   \
  do {                                                                \
    if ((act) == NBDT_ACT_SWISH) { MACRO(NBDT_ACT_SWISH); }           \
    else if ((act) == NBDT_ACT_RELU) { MACRO(NBDT_ACT_RELU); }        \
    else { MACRO(NBDT_ACT_NONE); }                                    \
  } while (0)


extern "C" int nbdt_bn_act_apply(const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                                 const float* beta, int32_t act, const float* gate, const void* residual, int32_t B,
                                 int32_t H, int32_t W, int32_t C, void* y, void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && y, "null argument");
  NBDT_REQUIRE(act >= 0 && act <= 2, "unknown activation");
  int rc = check_mb(B, H, W, C);
  if (rc) return rc;
  const MbGeom g = mb_geom(B, H, W, C, kPpt);
  const dim3 grid(g.slices, B), blk(g.threads);
  hipStream_t st = (hipStream_t)stream;
#define NBDT_GO(A)                                                                                              \
  do {                                                                                                          \
    if (gate && residual) hipLaunchKernelGGL((mb_apply_kernel<A, true, true>), grid, blk, 0, st, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, gate, (const bf16_t*)residual, g, kPpt, (bf16_t*)y); \
    else if (gate) hipLaunchKernelGGL((mb_apply_kernel<A, true, false>), grid, blk, 0, st, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, gate, (const bf16_t*)residual, g, kPpt, (bf16_t*)y); \
    else if (residual) hipLaunchKernelGGL((mb_apply_kernel<A, false, true>), grid, blk, 0, st, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, gate, (const bf16_t*)residual, g, kPpt, (bf16_t*)y); \
    else hipLaunchKernelGGL((mb_apply_kernel<A, false, false>), grid, blk, 0, st, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, gate, (const bf16_t*)residual, g, kPpt, (bf16_t*)y); \
  } while (0)
  NBDT_ACT_SWITCH(act, NBDT_GO);
#undef NBDT_GO
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_bn_act_pool(const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                                const float* beta, int32_t act, const void* mul, float scale, int32_t B, int32_t H,
                                int32_t W, int32_t C, float* out, void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && out, "null argument");
  NBDT_REQUIRE(act >= 0 && act <= 2, "unknown activation");
  int rc = check_mb(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  NBDT_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)B * C * sizeof(float), st));
  const MbGeom g = mb_geom(B, H, W, C, 2 * kPpt);
  const dim3 grid(g.slices, B), blk(g.threads);
  const size_t shmem = (size_t)g.threads * 8 * sizeof(float);
  float* target = out;           // deterministic mode: a zeroed copy of `out` per pixel slice, folded in slice order
  long long row_stride = 0;
  const size_t n_out = (size_t)B * C;
  if (deterministic() && g.slices > 1) {
    target = det_rows(st, (size_t)g.slices * n_out);
    if (!target) return nbdt::fail(NBDT_ENOMEM, "deterministic mode: %s (%s)", "no workspace for the per-slice rows", nbdt::det_rows_why());
    NBDT_HIP_CHECK(hipMemsetAsync(target, 0, (size_t)g.slices * n_out * sizeof(float), st));
    row_stride = (long long)n_out;
  }
#define NBDT_GO(A)                                                                                              \
  do {                                                                                                          \
    if (mul) hipLaunchKernelGGL((mb_pool_kernel<A, true>), grid, blk, shmem, st, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, (const bf16_t*)mul, g, 2 * kPpt, scale, target, row_stride); \
    else hipLaunchKernelGGL((mb_pool_kernel<A, false>), grid, blk, shmem, st, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, (const bf16_t*)mul, g, 2 * kPpt, scale, target, row_stride); \
  } while (0)
  NBDT_ACT_SWITCH(act, NBDT_GO);
#undef NBDT_GO
  NBDT_LAUNCH_CHECK();
  if (target != out) return det_fold(st, target, g.slices, n_out, out);
  return NBDT_OK;
}

static int bn_act_bwd_impl(const void* gu, const float* gate, const float* gpool, const void* x,
                           const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                           int32_t act, const void* gx_add, int32_t B, int32_t H, int32_t W, int32_t C,
                           float* scratch, float* dsum, float* dgamma, float* dbeta, void* gx, void* stream,
                           bool sums_ready, bool dsum_ready = false) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && (scratch || dsum_ready) && dsum && gx, "null argument");
  NBDT_REQUIRE(gu || (gpool && !gate), "need an upstream gradient tensor or a pooled gradient");
  NBDT_REQUIRE(!gate || (gu && gpool), "the SE form needs gu, gate and gpool");
  NBDT_REQUIRE(act >= 0 && act <= 2, "unknown activation");
  int rc = check_mb(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const bool se = gate != nullptr, pool = gu == nullptr;
  NBDT_REQUIRE(!(pool && gx_add), "pooled form has no gx_add");
  if (!sums_ready && !dsum_ready) {      // (sums_ready: the producing kernel -- nbdt_dwconv_bwd_data_bn -- already filled the slots)
    const MbGeom g = mb_geom(B, H, W, C, 2 * kPpt);
    const dim3 grid(g.slices, B), blk(g.threads);
    const size_t shmem = (size_t)g.threads * 16 * sizeof(float);
    SlotTarget tgt;
    const int nblk = g.slices * B;
    rc = slot_target(st, scratch, nblk, 2 * (size_t)C, &tgt);
    if (rc) return rc;
#define NBDT_GO(A)                                                                                              \
  do {                                                                                                          \
    if (se) hipLaunchKernelGGL((mb_bwd_reduce_kernel<A, true, false>), grid, blk, shmem, st, (const bf16_t*)gu, gate, gpool, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, 2 * kPpt, tgt.ptr, tgt.mask); \
    else if (pool) hipLaunchKernelGGL((mb_bwd_reduce_kernel<A, false, true>), grid, blk, shmem, st, (const bf16_t*)gu, gate, gpool, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, 2 * kPpt, tgt.ptr, tgt.mask); \
    else hipLaunchKernelGGL((mb_bwd_reduce_kernel<A, false, false>), grid, blk, shmem, st, (const bf16_t*)gu, gate, gpool, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, 2 * kPpt, tgt.ptr, tgt.mask); \
  } while (0)
    NBDT_ACT_SWITCH(act, NBDT_GO);
#undef NBDT_GO
    NBDT_LAUNCH_CHECK();
    rc = slot_finish(st, tgt, nblk, 2 * (size_t)C, scratch);
    if (rc) return rc;
  }
  if (!dsum_ready) {      // (dsum_ready: nbdt_bn_act_se_bwd_apply's own finalize kernel already wrote dsum / dgamma / dbeta)
    hipLaunchKernelGGL(mb_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, C, dsum, dgamma, dbeta);
    NBDT_LAUNCH_CHECK();
  }
  {
    const MbGeom g = mb_geom(B, H, W, C, kPpt);
    const dim3 grid(g.slices, B), blk(g.threads);
#define NBDT_GO(A)                                                                                              \
  do {                                                                                                          \
    if (se) { if (gx_add) hipLaunchKernelGGL((mb_bwd_apply_kernel<A, true, false, true>), grid, blk, 0, st, (const bf16_t*)gu, gate, gpool, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, dsum, (const bf16_t*)gx_add, g, kPpt, (bf16_t*)gx); \
              else hipLaunchKernelGGL((mb_bwd_apply_kernel<A, true, false, false>), grid, blk, 0, st, (const bf16_t*)gu, gate, gpool, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, dsum, (const bf16_t*)gx_add, g, kPpt, (bf16_t*)gx); } \
    else if (pool) hipLaunchKernelGGL((mb_bwd_apply_kernel<A, false, true, false>), grid, blk, 0, st, (const bf16_t*)gu, gate, gpool, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, dsum, (const bf16_t*)gx_add, g, kPpt, (bf16_t*)gx); \
    else { if (gx_add) hipLaunchKernelGGL((mb_bwd_apply_kernel<A, false, false, true>), grid, blk, 0, st, (const bf16_t*)gu, gate, gpool, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, dsum, (const bf16_t*)gx_add, g, kPpt, (bf16_t*)gx); \
           else hipLaunchKernelGGL((mb_bwd_apply_kernel<A, false, false, false>), grid, blk, 0, st, (const bf16_t*)gu, gate, gpool, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, dsum, (const bf16_t*)gx_add, g, kPpt, (bf16_t*)gx); } \
  } while (0)
    NBDT_ACT_SWITCH(act, NBDT_GO);
#undef NBDT_GO
    NBDT_LAUNCH_CHECK();
  }
  return NBDT_OK;
}

extern "C" int nbdt_bn_act_bwd(const void* gu, const float* gate, const float* gpool, const void* x,
                               const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                               int32_t act, const void* gx_add, int32_t B, int32_t H, int32_t W, int32_t C,
                               float* scratch, float* dsum, float* dgamma, float* dbeta, void* gx, void* stream) {
  return bn_act_bwd_impl(gu, gate, gpool, x, save_mean, save_rstd, gamma, beta, act, gx_add, B, H, W, C, scratch, dsum,
                         dgamma, dbeta, gx, stream, false);
}

extern "C" int nbdt_bn_act_bwd_apply(const void* gu, const void* x, const float* save_mean, const float* save_rstd,
                                     const float* gamma, const float* beta, int32_t act, const void* gx_add,
                                     int32_t B, int32_t H, int32_t W, int32_t C, float* scratch, float* dsum,
                                     float* dgamma, float* dbeta, void* gx, void* stream) {
  NBDT_REQUIRE(gu != nullptr, "null upstream gradient");
  return bn_act_bwd_impl(gu, nullptr, nullptr, x, save_mean, save_rstd, gamma, beta, act, gx_add, B, H, W, C, scratch,
                         dsum, dgamma, dbeta, gx, stream, true);
}

extern "C" int nbdt_bn_act_se_sums(const void* gu, const void* x, const float* save_mean, const float* save_rstd,
                                   const float* gamma, const float* beta, int32_t act, int32_t B, int32_t H, int32_t W,
                                   int32_t C, float* sums, void* stream) {
  NBDT_REQUIRE(gu && x && save_mean && save_rstd && gamma && beta && sums, "null argument");
  NBDT_REQUIRE(act >= 0 && act <= 2, "unknown activation");
  NBDT_REQUIRE(!deterministic(), "nbdt_bn_act_se_sums adds its pixel slices with atomics: use nbdt_bn_act_pool + nbdt_bn_act_bwd in deterministic mode");
  int rc = check_mb(B, H, W, C);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const MbGeom g = mb_geom(B, H, W, C, 2 * kPpt);
  const dim3 grid(g.slices, B), blk(g.threads);
  const size_t shmem = (size_t)g.threads * 40 * sizeof(float);
#define NBDT_GO(A)                                                                                              \
  hipLaunchKernelGGL((mb_se_sums_kernel<A>), grid, blk, shmem, st, (const bf16_t*)gu, (const bf16_t*)x, save_mean, save_rstd, gamma, beta, g, 2 * kPpt, sums)
  NBDT_ACT_SWITCH(act, NBDT_GO);
#undef NBDT_GO
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_bn_act_se_bwd_apply(const void* gu, const float* gate, const float* gpool, float* sums,
                                        const void* x, const float* save_mean, const float* save_rstd,
                                        const float* gamma, const float* beta, int32_t act, int32_t B, int32_t H,
                                        int32_t W, int32_t C, float* dsum, float* dgamma, float* dbeta, void* gx,
                                        void* stream) {
  NBDT_REQUIRE(gu && gate && gpool && sums && dsum, "null argument");
  int rc = check_mb(B, H, W, C);
  if (rc) return rc;
  hipLaunchKernelGGL(mb_se_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, sums, gate, gpool,
                     B, C, 1.f / (float)(H * W), dsum, dgamma, dbeta);
  NBDT_LAUNCH_CHECK();
  return bn_act_bwd_impl(gu, gate, gpool, x, save_mean, save_rstd, gamma, beta, act, nullptr, B, H, W, C, nullptr, dsum,
                         dgamma, dbeta, gx, stream, false, true);
}

static int check_dw(int B, int H, int W, int C, int k, int stride) {
  int rc = check_mb(B, H, W, C);
  if (rc) return rc;
  NBDT_REQUIRE(k == 3 || k == 5, "depthwise kernel size must be 3 or 5");
  NBDT_REQUIRE(stride == 1 || stride == 2, "stride must be 1 or 2");
  NBDT_REQUIRE(H % stride == 0 && W % stride == 0, "spatial size must be divisible by the stride");
  return NBDT_OK;
}

static DwGeom dw_geom(int B, int H, int W, int C, int k, int stride, int ppt) {
  DwGeom d;
  d.out = mb_geom(B, H / stride, W / stride, C, ppt);
  d.Hi = H; d.Wi = W;
  d.rowi = (W + 2) * C;
  d.imgi = (H + 2) * d.rowi;
  d.k = k; d.stride = stride; d.pad = k / 2;
  return d;
}

extern "C" int nbdt_dwconv_fwd(const void* x, const float* w, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                               int32_t stride, void* y, float* bn_scratch, void* stream) {
  NBDT_REQUIRE(x && w && y, "null argument");
  int rc = check_dw(B, H, W, C, k, stride);
  if (rc) return rc;
  const DwGeom d = dw_geom(B, H, W, C, k, stride, 4);
  hipStream_t st = (hipStream_t)stream;
  if (k == 3) return stride == 1 ? launch_dw_row<3, 1, false>(x, w, d, B, y, bn_scratch, st)
                                 : launch_dw_row<3, 2, false>(x, w, d, B, y, bn_scratch, st);
  return stride == 1 ? launch_dw_row<5, 1, false>(x, w, d, B, y, bn_scratch, st)
                     : launch_dw_row<5, 2, false>(x, w, d, B, y, bn_scratch, st);
}

extern "C" int nbdt_dwconv_bwd_data(const void* gy, const float* w, int32_t B, int32_t H, int32_t W, int32_t C,
                                    int32_t k, int32_t stride, void* gx, void* stream) {
  NBDT_REQUIRE(gy && w && gx, "null argument");
  int rc = check_dw(B, H, W, C, k, stride);
  if (rc) return rc;
  const MbGeom in = mb_geom(B, H, W, C, 4);
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) {   // same correlation with the kernel rotated by 180 degrees
    const DwGeom d = dw_geom(B, H, W, C, k, 1, 4);
    return k == 3 ? launch_dw_row<3, 1, true>(gy, w, d, B, gx, nullptr, st)
                  : launch_dw_row<5, 1, true>(gy, w, d, B, gx, nullptr, st);
  } else if (W % 4 == 0) {     // row-segment form: segments of 8 (or 4) input pixels
    const int tw = W % 8 == 0 ? 8 : 4;
    const int nseg = W / tw;
    int PY = kThreads / in.c8;
    if (PY < 1) PY = 1;
    if (PY > H * nseg) PY = H * nseg;
    const dim3 grid((H * nseg + PY - 1) / PY, B), blk(in.c8 * PY);
#define NBDT_GO(K, TW) hipLaunchKernelGGL((dw_bwd_data_s2_row_kernel<K, TW>), grid, blk, 0, st, (const bf16_t*)gy, w, in, H / 2, W / 2, nseg, PY, (bf16_t*)gx)
    if (k == 3) { if (tw == 8) NBDT_GO(3, 8); else NBDT_GO(3, 4); }
    else        { if (tw == 8) NBDT_GO(5, 8); else NBDT_GO(5, 4); }
#undef NBDT_GO
  } else {
    hipLaunchKernelGGL(dw_bwd_data_kernel, dim3(in.slices, B), dim3(in.threads), 0, st, (const bf16_t*)gy, w, in,
                       H / stride, W / stride, k, stride, k / 2, 4, (bf16_t*)gx);
  }
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_dwconv_bwd_data_bn(const void* gy, const float* w, int32_t B, int32_t H, int32_t W, int32_t C,
                                      int32_t k, void* gx, const void* bn_x, const float* save_mean,
                                      const float* save_rstd, const float* gamma, const float* beta, float* scratch,
                                      void* stream) {
  NBDT_REQUIRE(gy && w && gx && bn_x && save_mean && save_rstd && gamma && beta && scratch, "null argument");
  int rc = check_dw(B, H, W, C, k, 1);
  if (rc) return rc;
  const DwGeom d = dw_geom(B, H, W, C, k, 1, 4);
  const DwBn bn{(const bf16_t*)bn_x, save_mean, save_rstd, gamma, beta};
  hipStream_t st = (hipStream_t)stream;
  return k == 3 ? launch_dw_row<3, 1, true>(gy, w, d, B, gx, scratch, st, &bn)
                : launch_dw_row<5, 1, true>(gy, w, d, B, gx, scratch, st, &bn);
}

extern "C" int nbdt_dwconv_bwd_weight(const void* x, const void* gy, int32_t B, int32_t H, int32_t W, int32_t C,
                                      int32_t k, int32_t stride, float* dw, void* stream) {
  NBDT_REQUIRE(x && gy && dw, "null argument");
  int rc = check_dw(B, H, W, C, k, stride);
  if (rc) return rc;
  const DwGeom d = dw_geom(B, H, W, C, k, stride, 1);
  const int Ho = H / stride;
  const int cblocks = (d.out.c8 + 31) / 32, CB = (d.out.c8 + cblocks - 1) / cblocks;
  int PY = kThreads / CB;
  if (PY > Ho) PY = Ho;
  const int threads = CB * PY;
  // every block ends in CB*k*8 global atomics on addresses every other block of its channels adds to as well: let a
  // thread walk about NBDT_DW_TARGET output pixels (several images when rows are short) before the block folds
  // (profiles/r04_dw_wgrad.txt: rows of 28+ outputs are fastest at ~128 pixels per thread, shorter ones at ~64)
#ifdef NBDT_DW_TARGET            // timing-only builds: one target everywhere
  const int target_px = NBDT_DW_TARGET;
#else
  const int target_px = W / stride >= 28 ? 128 : 64;
#endif
  int bchunk = target_px / (W / stride);
  if (bchunk < 1) bchunk = 1;
  if (bchunk > B) bchunk = B;
  const dim3 grid(((Ho + PY - 1) / PY) * cblocks, (B + bchunk - 1) / bchunk, k), blk(threads);
  const size_t shmem = (size_t)threads * k * 8 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  float* target = dw;
  long long row_stride = 0;
  const size_t n_dw = (size_t)k * k * C;
  const int nrows = (int)(grid.x * grid.y);
  if (deterministic() && nrows > 1) {
    target = det_rows(st, (size_t)nrows * n_dw);
    if (!target) return nbdt::fail(NBDT_ENOMEM, "deterministic mode: %s (%s)", "no workspace for the per-block rows", nbdt::det_rows_why());
    NBDT_HIP_CHECK(hipMemsetAsync(target, 0, (size_t)nrows * n_dw * sizeof(float), st));
    row_stride = (long long)n_dw;
  }
#define NBDT_GO(K, S, U) hipLaunchKernelGGL((dw_bwd_weight_kernel<K, S, U>), grid, blk, shmem, st, (const bf16_t*)x, (const bf16_t*)gy, d, PY, CB, cblocks, bchunk, target, row_stride)
#ifdef NBDT_DW_U          // timing-only builds: one U everywhere
#define NBDT_GO_S1(K) { NBDT_GO(K, 1, NBDT_DW_U); }
#define NBDT_GO_S2(K) { NBDT_GO(K, 2, NBDT_DW_U); }
#else
  // pixels in flight per thread: the short-row layers (<= 28 wide: few waves per CU, a row is a chain of round trips)
  // want 8; the long-row stride-1 layers are bandwidth-bound and lose occupancy to the registers of more than 1
  const bool long_rows = W / stride >= 56;
#define NBDT_GO_S1(K) { if (long_rows) NBDT_GO(K, 1, 1); else NBDT_GO(K, 1, 8); }
#define NBDT_GO_S2(K) { NBDT_GO(K, 2, 4); }
#endif
  if (k == 3) { if (stride == 1) NBDT_GO_S1(3) else NBDT_GO_S2(3) }
  else { if (stride == 1) NBDT_GO_S1(5) else NBDT_GO_S2(5) }
#undef NBDT_GO_S1
#undef NBDT_GO_S2
#undef NBDT_GO
  NBDT_LAUNCH_CHECK();
  if (target != dw) return det_fold(st, target, nrows, n_dw, dw);
  return NBDT_OK;
}

extern "C" int nbdt_se_gate_fwd(const float* pooled, const float* w1, const float* b1, const float* w2,
                                const float* b2, int32_t B, int32_t C, int32_t C_real, int32_t S, float* pre1,
                                float* gate, void* stream) {
  NBDT_REQUIRE(pooled && w1 && b1 && w2 && b2 && pre1 && gate, "null argument");
  NBDT_REQUIRE(B > 0 && C_real > 0 && C_real <= C && S > 0 && S <= 256, "bad SE sizes");
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3(B), dim3(C_real > 256 ? 1024 : 256),
                     (size_t)(C_real + S) * sizeof(float), (hipStream_t)stream, pooled, w1, b1, w2, b2, C, C_real, S,
                     pre1, gate);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_se_gate_bwd(const float* dgate, const float* gate, const float* pre1, const float* pooled,
                                const float* w1, const float* w2, int32_t B, int32_t C, int32_t C_real, int32_t S,
                                float* dpre2, float* dpre1, float* gpool, float* dw1, float* db1, float* dw2,
                                float* db2, void* stream) {
  NBDT_REQUIRE(dgate && gate && pre1 && pooled && w1 && w2 && dpre2 && dpre1 && gpool, "null argument");
  NBDT_REQUIRE((dw1 && db1 && dw2 && db2) || (!dw1 && !db1 && !dw2 && !db2), "parameter gradients: all four or none");
  NBDT_REQUIRE(B > 0 && C_real > 0 && C_real <= C && S > 0 && S <= 256, "bad SE sizes");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_gate_bwd_kernel, dim3(B), dim3(C_real > 256 ? 1024 : 256),
                     (size_t)(C_real + S + (C_real > 256 ? 16 : 4) * S) * sizeof(float), st, dgate, gate, pre1, w1, w2, C, C_real, S, dpre2, dpre1,
                     gpool);
  NBDT_LAUNCH_CHECK();
  if (!dw1) return NBDT_OK;        // data part only: the caller runs nbdt_se_param_grad where it likes (another stream)
  return nbdt_se_param_grad(dpre2, dpre1, pre1, pooled, B, C, C_real, S, dw1, db1, dw2, db2, stream);
}

extern "C" int nbdt_se_param_grad(const float* dpre2, const float* dpre1, const float* pre1, const float* pooled,
                                  int32_t B, int32_t C, int32_t C_real, int32_t S, float* dw1, float* db1, float* dw2,
                                  float* db2, void* stream) {
  NBDT_REQUIRE(dpre2 && dpre1 && pre1 && pooled && dw1 && db1 && dw2 && db2, "null argument");
  NBDT_REQUIRE(B > 0 && C_real > 0 && C_real <= C && S > 0 && S <= 256, "bad SE sizes");
  hipStream_t st = (hipStream_t)stream;
  const int n = C_real * S;
  const int bchunk = deterministic() ? B : 16;     // (one chunk: one add per address)
  hipLaunchKernelGGL(se_param_grad_kernel, dim3((n + 255) / 256, (B + bchunk - 1) / bchunk), dim3(256), 0, st, dpre2,
                     dpre1, pre1, pooled, B, C, C_real, S, bchunk, dw1, db1, dw2, db2);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_dropout_fwd(const float* x, int64_t n, float p, uint32_t seed, uint8_t* mask, float* y,
                                void* stream) {
  NBDT_REQUIRE(x && mask && y, "null argument");
  NBDT_REQUIRE(n > 0 && p >= 0.f && p < 1.f, "bad dropout arguments");
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)n, p, 1.f / (1.f - p), seed, mask, y);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_dropout_bwd(const float* gy, int64_t n, float p, const uint8_t* mask, float* gx, void* stream) {
  NBDT_REQUIRE(gy && mask && gx, "null argument");
  NBDT_REQUIRE(n > 0 && p >= 0.f && p < 1.f, "bad dropout arguments");
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gy,
                     (long long)n, 1.f / (1.f - p), mask, gx);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}
