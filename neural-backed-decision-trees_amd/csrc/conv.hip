// Implicit-GEMM convolution on MFMA (gfx950), bf16 operands / fp32 accumulate.
//
// Replaces cuDNN/MIOpen behind nn.Conv2d for every conv on the NBDT backbone path
// (reference nbdt/models/resnet.py:47-66 BasicBlock convs and shortcut, pytorchcv WRN PreResUnit
// convs used by nbdt/models/wideresnet.py:1-5) -- forward, data-gradient and 1x1 shortcut are all
// the same kernel driven by a "tap table" (nbdt_conv_desc in include/nbdt_hip.h):
//
//     out[pix(m)][n] (+)= sum_t sum_c  in[pix_in(m) + tap_off[t]][c] * w[n][w_tap[t]][c]
//
// Layout decisions (MI355X-first):
//   * activations are padded NHWC bf16, so each tap of each pixel is `cin` CONTIGUOUS bf16 and
//     never out of bounds: the K loop is (tap, 32-channel chunk) with no predication at all;
//   * weights are [cout][tap][cin] bf16 so both MFMA operands are K-contiguous and every LDS
//     fragment read is one ds_read_b128;
//   * block tile = 256 pixels x (32*NT) couts x 32 k; 4 waves, wave w owns pixels [64w, 64w+64)
//     and ALL couts of the tile: acc[NT][2] tiles of v_mfma_f32_32x32x16_bf16 (NT=5 -> 160 couts,
//     which divides every WideResNet-28-10 width 160/320/640; NT=4/2/1 for power-of-two widths);
//   * MFMA is issued "transposed" (A operand = weights, B operand = pixels) so each lane ends up
//     with 4 consecutive couts of ONE pixel per accumulator quad -> 8-byte bf16 stores;
//   * LDS tiles are [row][32 k] (64 B rows) with the 16-B chunk index XOR-swizzled by
//     (row>>2)&3: conflict-free for the ds_read_b128 lane groups of a 32-row fragment and for
//     the ds_write_b128 staging writes (derivation in DESIGN.md);
//   * global->register->LDS staging, double-buffered, ONE barrier per K tile: the loads of tile
//     t+1 are issued before the MFMAs of tile t and written to the other buffer after them;
//   * 1-D grid remapped so each XCD walks a contiguous range of (pixel-tile, cout-tile) items
//     with the cout-tile fastest: tiles that share input rows / halos hit the same private L2.
//
// Roofline: MFMA-bound.  flops = 2*M*cout*ntaps*cin per launch.
#include "common.h"
#include <stdlib.h>

using namespace nbdt;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct ConvParams {
  nbdt_conv_desc d;
  const bf16_t* in;
  const bf16_t* w;
  bf16_t* out;
  const bf16_t* res;  // residual (or out itself when accumulating); may be null
  int M;              // B*gh*gw
  int n_blocks;       // cout / (32*NT)
  int m_blocks;       // ceil(M/256)
  int per_xcd;        // ceil(m_blocks*n_blocks / 8)
};

constexpr int BM = 256;
constexpr int BK = 32;

// byte offset of (row, 16-byte chunk c) inside a [rows][32] bf16 LDS tile
__device__ __forceinline__ int lds_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ int pix_offset(int m, int gh, int gw, int bs, int hs, int ws, int base) {
  const int j = m % gw;
  const int t = m / gw;
  const int i = t % gh;
  const int b = t / gh;
  return b * bs + i * hs + j * ws + base;
}

template <int NT, bool HAS_RES>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvParams p) {
  constexpr int BN = 32 * NT;
  constexpr int A_BYTES = BM * BK * 2;  // 16 KiB
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int W_ITERS = (BN * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [A buf0][A buf1][W buf0][W buf1]

  // ---- XCD-aware work item (see header)
  const int bid = blockIdx.x;
  const int item = (bid & 7) * p.per_xcd + (bid >> 3);
  if (item >= p.m_blocks * p.n_blocks) return;
  const int m_blk = item / p.n_blocks;
  const int n_blk = item - m_blk * p.n_blocks;
  const int m0 = m_blk * BM;
  const int n0 = n_blk * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const nbdt_conv_desc& d = p.d;
  const int kchunks = d.cin >> 5;
  const int nk = d.ntaps * kchunks;
  const int w_row_len = d.w_ntaps * d.cin;

  // ---- staging assignments
  const int a_chunk = tid & 3;
  int a_goff[4];  // element offset of this thread's 4 pixel rows (+ its 8-element chunk)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + (tid >> 2) + 64 * i;
    m = m < p.M ? m : p.M - 1;
    a_goff[i] = pix_offset(m, d.gh, d.gw, d.in_bs, d.in_hs, d.in_ws, d.in_base) + a_chunk * 8;
  }
  int w_goff[W_ITERS];
#pragma unroll
  for (int i = 0; i < W_ITERS; ++i) {
    const int idx = tid + 256 * i;
    w_goff[i] = (n0 + (idx >> 2)) * w_row_len + (idx & 3) * 8;
  }

  u32x4 ra[4];
  u32x4 rw[W_ITERS];

  auto load_tile = [&](int tap, int kc) {
    const int a_k = d.tap_off[tap] + kc * BK;
    const int w_k = d.w_tap[tap] * d.cin + kc * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *(const u32x4*)(p.in + (a_goff[i] + a_k));
#pragma unroll
    for (int i = 0; i < W_ITERS; ++i)
      if (tid + 256 * i < BN * 4) rw[i] = *(const u32x4*)(p.w + (w_goff[i] + w_k));
  };
  auto store_tile = [&](int buf) {
    unsigned char* As = smem + buf * A_BYTES;
    unsigned char* Ws = smem + 2 * A_BYTES + buf * W_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(As + lds_off((tid >> 2) + 64 * i, a_chunk)) = ra[i];
#pragma unroll
    for (int i = 0; i < W_ITERS; ++i) {
      const int idx = tid + 256 * i;
      if (idx < BN * 4) *(u32x4*)(Ws + lds_off(idx >> 2, idx & 3)) = rw[i];
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

  auto compute = [&](int buf) {
    const unsigned char* As = smem + buf * A_BYTES;
    const unsigned char* Ws = smem + 2 * A_BYTES + buf * W_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 2 * ks + frag_half;
      bf16x8 pf[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
        pf[tm] = *(const bf16x8*)(As + lds_off(wave * 64 + tm * 32 + frag_row, c));
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const bf16x8 wf = *(const bf16x8*)(Ws + lds_off(tn * 32 + frag_row, c));
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, pf[tm], acc[tn][tm], 0, 0, 0);
      }
    }
  };

  // ---- main loop: one barrier per K tile
  int tap = 0, kc = 0;
  load_tile(0, 0);
  store_tile(0);
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    // tap is the FAST loop index: a 32-channel slice of the block's input rows (+halo) is 64 B per
    // pixel, so the 9 shifted re-reads of it stay inside the XCD's 4 MiB L2 (tap-outer order kept
    // the whole 320-B pixel live across taps: 7 MB per XCD, measured 3-6x over-fetch from HBM/MALL)
    if (++tap == d.ntaps) { tap = 0; ++kc; }
    const bool more = it + 1 < nk;
    if (more) load_tile(tap, kc);
    compute(cur);
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns pixel (lane&31) of each of its 2 pixel tiles; per accumulator quad
  // q it holds couts 8q + 4*(lane>>5) .. +3 of cout-tile tn  ->  one 8-byte store
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wave * 64 + tm * 32 + frag_row;
    if (m >= p.M) continue;
    const int o = pix_offset(m, d.gh, d.gw, d.out_bs, d.out_hs, d.out_ws, d.out_base) + n0 + 4 * frag_half;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      u32x2 rr[4];
      if (HAS_RES) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rr[q] = *(const u32x2*)(p.res + (o + tn * 32 + q * 8));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v0 = acc[tn][tm][4 * q + 0], v1 = acc[tn][tm][4 * q + 1];
        float v2 = acc[tn][tm][4 * q + 2], v3 = acc[tn][tm][4 * q + 3];
        if (HAS_RES) {
          v0 += __uint_as_float(rr[q][0] << 16);
          v1 += __uint_as_float(rr[q][0] & 0xffff0000u);
          v2 += __uint_as_float(rr[q][1] << 16);
          v3 += __uint_as_float(rr[q][1] & 0xffff0000u);
        }
        u32x2 pk;
        pk[0] = pack_bf16x2(v0, v1);
        pk[1] = pack_bf16x2(v2, v3);
        *(u32x2*)(p.out + (o + tn * 32 + q * 8)) = pk;
      }
    }
  }
}

template <int NT>
static int launch(ConvParams& p, hipStream_t st) {
  constexpr int BN = 32 * NT;
  p.n_blocks = p.d.cout / BN;
  p.m_blocks = (p.M + BM - 1) / BM;
  const int items = p.m_blocks * p.n_blocks;
  p.per_xcd = (items + 7) / 8;
  const size_t shmem = 2 * (BM * BK * 2) + 2 * (BN * BK * 2);
  if (p.res != nullptr)
    hipLaunchKernelGGL((conv_igemm_kernel<NT, true>), dim3(p.per_xcd * 8), dim3(256), shmem, st, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<NT, false>), dim3(p.per_xcd * 8), dim3(256), shmem, st, p);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

static int conv_igemm_impl(const nbdt_conv_desc* d, const void* in, const void* w, void* out, const void* residual,
                           float* bn_scratch, void* stream, const nbdt::BnBwdArgs* bn = nullptr) {
  NBDT_REQUIRE(d && in && w && out, "null argument");
  NBDT_REQUIRE(d->cin > 0 && d->cin % 32 == 0, "cin must be a multiple of 32");
  NBDT_REQUIRE(d->cout > 0 && d->cout % 32 == 0, "cout must be a multiple of 32");
  NBDT_REQUIRE(d->ntaps >= 1 && d->ntaps <= 9 && d->w_ntaps >= 1, "bad tap table");
  for (int t = 0; t < d->ntaps; ++t) NBDT_REQUIRE(d->w_tap[t] >= 0 && d->w_tap[t] < d->w_ntaps, "bad w_tap");
  NBDT_REQUIRE(d->B > 0 && d->gh > 0 && d->gw > 0, "empty pixel grid");
  NBDT_REQUIRE((d->in_base % 8) == 0 && (d->in_ws % 8) == 0 && (d->in_hs % 8) == 0 && (d->in_bs % 8) == 0,
               "input pixel offsets must be 16-byte aligned");
  NBDT_REQUIRE((d->out_base % 4) == 0 && (d->out_ws % 4) == 0 && (d->out_hs % 4) == 0 && (d->out_bs % 4) == 0,
               "output pixel offsets must be 8-byte aligned");
  for (int t = 0; t < d->ntaps; ++t) NBDT_REQUIRE(d->tap_off[t] % 8 == 0, "tap offsets must be 16-byte aligned");
  ConvParams p;
  p.d = *d;
  p.in = (const bf16_t*)in;
  p.w = (const bf16_t*)w;
  p.out = (bf16_t*)out;
  p.res = d->accumulate ? (const bf16_t*)out : (const bf16_t*)residual;
  NBDT_REQUIRE(!(d->accumulate && residual), "accumulate and residual are exclusive");
  const int64_t M64 = (int64_t)d->B * d->gh * d->gw;
  NBDT_REQUIRE(M64 < (1ll << 31), "pixel grid too large");
  p.M = (int)M64;
  hipStream_t st = (hipStream_t)stream;
  // default: v2 (LDS-DMA 3-stage pipeline, conv_dma.hip); NBDT_IGEMM_V1=1 keeps the register-staged
  // kernel below for A/B measurements
  static const bool use_v1 = getenv("NBDT_IGEMM_V1") != nullptr;
  if (!use_v1) {
    // dense 3x3 / stride-1 convs whose pixel tiles are whole rows / images: LDS-resident halo tile
    static const bool no_halo = getenv("NBDT_NO_HALO") != nullptr;
    nbdt::HaloGeom hg;
    if (!no_halo && nbdt::conv_halo_applicable(d, p.M, &hg))
      return nbdt::conv3x3_halo(d, hg, in, w, out, p.res, bn_scratch, bn, p.M, st);
    return nbdt::conv_igemm_dma(d, in, w, out, p.res, bn_scratch, bn, p.M, st);
  }
  NBDT_REQUIRE(bn_scratch == nullptr, "fused BN statistics need the LDS-DMA kernel (unset NBDT_IGEMM_V1)");
  const int nt32 = d->cout / 32;
  if (nt32 % 5 == 0) return launch<5>(p, st);
  if (nt32 % 4 == 0) return launch<4>(p, st);
  if (nt32 % 2 == 0) return launch<2>(p, st);
  return launch<1>(p, st);
}

extern "C" int nbdt_conv_igemm(const nbdt_conv_desc* d, const void* in, const void* w, void* out,
                               const void* residual, void* stream) {
  return conv_igemm_impl(d, in, w, out, residual, nullptr, stream);
}

extern "C" int nbdt_conv_igemm_stats(const nbdt_conv_desc* d, const void* in, const void* w, void* out,
                                     const void* residual, float* bn_scratch, void* stream) {
  NBDT_REQUIRE(bn_scratch != nullptr, "null statistics workspace");
  NBDT_REQUIRE(d && !d->accumulate, "fused statistics are for plain outputs");
  return conv_igemm_impl(d, in, w, out, residual, bn_scratch, stream);
}

extern "C" int nbdt_conv_igemm_bnbwd(const nbdt_conv_desc* d, const void* in, const void* w, void* out,
                                     const void* bn_x, const float* save_mean, const float* save_rstd,
                                     const float* gamma, const float* beta, float* bn_partials, void* stream) {
  NBDT_REQUIRE(bn_x && save_mean && save_rstd && gamma && beta && bn_partials, "null BatchNorm argument");
  NBDT_REQUIRE(d && !d->accumulate, "fused BatchNorm-backward sums are for plain outputs");
  nbdt::BnBwdArgs bn{bn_x, save_mean, save_rstd, gamma, beta};
  return conv_igemm_impl(d, in, w, out, nullptr, bn_partials, stream, &bn);
}

extern "C" int nbdt_conv_igemm_affine(const nbdt_conv_desc* d, const void* in, const void* w, void* out,
                                      const void* residual, const float* scale, const float* shift, int32_t act,
                                      void* stream) {
  NBDT_REQUIRE(scale && shift, "null scale / shift");
  NBDT_REQUIRE(act >= 0 && act <= 2, "unknown activation");
  NBDT_REQUIRE(d && !d->accumulate, "the fused inference epilogue is for plain outputs");
  static const bool v1 = getenv("NBDT_IGEMM_V1") != nullptr;
  NBDT_REQUIRE(!v1, "the fused inference epilogue needs the LDS-DMA kernels (unset NBDT_IGEMM_V1)");
  nbdt::BnBwdArgs ep{nullptr, nullptr, nullptr, nullptr, nullptr};
  ep.aff_scale = scale; ep.aff_shift = shift; ep.aff_act = act;
  return conv_igemm_impl(d, in, w, out, residual, nullptr, stream, &ep);
}

// ------------------------------------------------------------------------------------------
// weight prep: fp32 master [cout][taps][cin] -> bf16 same order (+ optional dgrad copy
// wd[cin][taps][cout], tap order reversed).  Tiny, memory-bound.
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ w, int cout, int taps,
                                                          int cin, bf16_t* __restrict__ wb,
                                                          bf16_t* __restrict__ wd) {
  const int64_t n = (int64_t)cout * taps * cin;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const bf16_t v = f32_to_bf16(w[i]);
    if (wb) wb[i] = v;
    if (wd) {
      const int ci = (int)(i % cin);
      const int64_t r = i / cin;
      const int t = (int)(r % taps);
      const int co = (int)(r / taps);
      wd[((int64_t)ci * taps + (taps - 1 - t)) * cout + co] = v;
    }
  }
}

// All conv layers in ONE launch (the per-layer form costs 30 launches x 17 us per optimizer step).
// Work item = one 64(cout) x 32(cin) tile of one tap of one layer, transposed through LDS so the fp32 reads
// (128 B along cin) and the bf16 writes (64 couts = 128 B, one full line) are both coalesced.  (An element-wise
// version with 2-byte scattered writes took 556 us for WRN-28-10; 32-cout tiles wrote half lines and the PMC
// pass showed 1.1 GB of HBM writes for a 73 MB output.)  table[l] = {src element offset in the flat fp32 buffer,
// dst element offset in wd_flat, cout, taps, cin, first tile index of the layer}; tiles per layer =
// taps * ceil(cout/64) * (cin/32).
__global__ __launch_bounds__(256) void weight_prep_batched_kernel(const float* __restrict__ flat,
                                                                  const long long* __restrict__ table, int n_layers,
                                                                  long long total_tiles, bf16_t* __restrict__ wd_flat) {
  __shared__ long long tab[64 * 6];
  __shared__ float tile[64][33];
  for (int i = threadIdx.x; i < n_layers * 6; i += 256) tab[i] = table[i];
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  const int wx = threadIdx.x & 63, wy = threadIdx.x >> 6;      // 64 x 4
  for (long long e = blockIdx.x; e < total_tiles; e += gridDim.x) {
    int l = 0;
    while (l + 1 < n_layers && e >= tab[(l + 1) * 6 + 5]) ++l;
    const long long* T = tab + l * 6;
    const int cout = (int)T[2], taps = (int)T[3], cin = (int)T[4];
    const int cit = cin >> 5, cot = (cout + 63) >> 6;
    long long i = e - T[5];
    const int ci0 = (int)(i % cit) * 32;
    i /= cit;
    const int co0 = (int)(i % cot) * 64;
    const int t = (int)(i / cot);
    const float* src = flat + T[0];
    bf16_t* dst = wd_flat + T[1];
#pragma unroll
    for (int r = ty; r < 64; r += 8)
      tile[r][tx] = co0 + r < cout ? src[((long long)(co0 + r) * taps + t) * cin + ci0 + tx] : 0.f;
    __syncthreads();
#pragma unroll
    for (int r = wy; r < 32; r += 4)
      if (co0 + wx < cout)
        dst[((long long)(ci0 + r) * taps + (taps - 1 - t)) * cout + co0 + wx] = f32_to_bf16(tile[wx][r]);
    __syncthreads();
  }
}

extern "C" int nbdt_weight_prep_batched(const float* flat, const int64_t* table, int32_t n_layers, int64_t total_tiles,
                                        void* wd_flat, void* stream) {
  NBDT_REQUIRE(flat && table && wd_flat, "null argument");
  NBDT_REQUIRE(n_layers > 0 && n_layers <= 64 && total_tiles > 0, "bad layer table");
  long long blocks = total_tiles < 8192 ? total_tiles : 8192;
  hipLaunchKernelGGL(weight_prep_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flat,
                     (const long long*)table, n_layers, (long long)total_tiles, (bf16_t*)wd_flat);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

__global__ __launch_bounds__(256) void weight_tile_batched_kernel(const bf16_t* __restrict__ src,
                                                                  const long long* __restrict__ table, int n,
                                                                  long long total_tiles, bf16_t* __restrict__ dst) {
  __shared__ long long tab[64 * 5];
  for (int i = threadIdx.x; i < n * 5; i += 256) tab[i] = table[i];
  __syncthreads();
  for (long long e = blockIdx.x; e < total_tiles; e += gridDim.x) {
    int l = 0;
    while (l + 1 < n && e >= tab[(l + 1) * 5 + 4]) ++l;
    const long long* T = tab + l * 5;
    const int rows = (int)T[2], k = (int)T[3];
    const int nt32 = rows / 32;
    const int nt = nt32 % 5 == 0 ? 5 : (nt32 % 4 == 0 ? 4 : (nt32 % 2 == 0 ? 2 : 1));
    const int bn = 32 * nt, kchunks = k / 32;
    long long i = e - T[4];                       // tile index = (n_blk * kchunks + kc) * 9 + tap
    const int tap = (int)(i % 9);
    i /= 9;
    const int kc = (int)(i % kchunks);
    const int n_blk = (int)(i / kchunks);
    const bf16_t* s0 = src + T[0];
    bf16_t* d0 = dst + T[1] + (e - T[4]) * (long long)(bn * 32);
    for (int q = threadIdx.x; q < bn * 4; q += 256) {     // (row, LDS chunk position)
      const int r = q >> 2, cp = q & 3;
      const int c = cp ^ ((r >> 2) & 3);
      const u32x4_t v = *(const u32x4_t*)(s0 + ((long long)(n_blk * bn + r) * 9 + tap) * k + kc * 32 + c * 8);
      *(u32x4_t*)(d0 + r * 32 + cp * 8) = v;
    }
  }
}

extern "C" int nbdt_weight_tile_batched(const void* src_bf16, const int64_t* table, int32_t n, int64_t total_tiles,
                                        void* dst_bf16, void* stream) {
  NBDT_REQUIRE(src_bf16 && table && dst_bf16, "null argument");
  NBDT_REQUIRE(n > 0 && n <= 64 && total_tiles > 0, "bad matrix table");
  long long blocks = total_tiles < 8192 ? total_tiles : 8192;
  hipLaunchKernelGGL(weight_tile_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src_bf16, (const long long*)table, n, (long long)total_tiles, (bf16_t*)dst_bf16);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_weight_prep(const float* w, int32_t cout, int32_t taps, int32_t cin, void* w_bf16,
                                void* wd_bf16, void* stream) {
  NBDT_REQUIRE(w && (w_bf16 || wd_bf16), "null argument");
  NBDT_REQUIRE(cout > 0 && taps > 0 && cin > 0, "bad shape");
  const int64_t n = (int64_t)cout * taps * cin;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(weight_prep_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, cout, taps, cin,
                     (bf16_t*)w_bf16, (bf16_t*)wd_bf16);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}
