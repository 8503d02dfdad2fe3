// Implicit-GEMM convolution on MFMA (gfx950), bf16 operands / fp32 accumulate: C-ABI entry points and dispatch.
//
// Replaces cuDNN/MIOpen behind nn.Conv2d for every conv on the NBDT backbone path
// (reference nbdt/models/resnet.py:47-66 BasicBlock convs and shortcut, pytorchcv WRN PreResUnit
// convs used by nbdt/models/wideresnet.py:1-5) -- forward, data-gradient and 1x1 shortcut are all
// the same computation driven by a "tap table" (nbdt_conv_desc in include/nbdt_hip.h):
//
//     out[pix(m)][n] (+)= sum_t sum_c  in[pix_in(m) + tap_off[t]][c] * w[n][w_tap[t]][c]
//
// Layout decisions (MI355X-first):
//   * activations are padded NHWC bf16, so each tap of each pixel is `cin` CONTIGUOUS bf16 and
//     never out of bounds: the K loop is (tap, 32-channel chunk) with no predication at all;
//   * weights are [cout][tap][cin] bf16 so both MFMA operands are K-contiguous and every LDS
//     fragment read is one ds_read_b128;
//   * a wave owns 64 pixels x ALL couts of the block tile: acc[NT][2] tiles of v_mfma_f32_32x32x16_bf16
//     (NT=5 -> 160 couts, which divides every WideResNet-28-10 width 160/320/640; NT=4/2/1 otherwise);
//   * MFMA is issued "transposed" (A operand = weights, B operand = pixels) so each lane ends up
//     with 4 consecutive couts of ONE pixel per accumulator quad;
//   * LDS tiles are [row][32 k] (64 B rows) with the 16-B chunk index XOR-swizzled by (row>>2)&3:
//     conflict-free for the ds_read_b128 lane groups of a 32-row fragment (derivation in DESIGN.md);
//   * 1-D grid remapped so each XCD walks a contiguous range of (pixel-tile, cout-tile) items.
//
// Kernels: conv_halo.hip (dense 3x3 stride-1: 97 % of the backbone's igemm flops) and conv_dma.hip (every other
// shape).  Roofline: MFMA-bound.  flops = 2*M*cout*ntaps*cin per launch.
#include "common.h"
#include <stdlib.h>

using namespace nbdt;

static int check_desc(const nbdt_conv_desc* d);

static int conv_igemm_impl(const nbdt_conv_desc* d, const void* in, const void* w, void* out, const void* residual,
                           float* bn_scratch, void* stream, const nbdt::BnBwdArgs* bn = nullptr) {
  NBDT_REQUIRE(d && in && w && out, "null argument");
  int rc = check_desc(d);
  if (rc) return rc;
  const void* res = d->accumulate ? (const void*)out : residual;
  NBDT_REQUIRE(!(d->accumulate && residual), "accumulate and residual are exclusive");
  const int64_t M64 = (int64_t)d->B * d->gh * d->gw;
  NBDT_REQUIRE(M64 < (1ll << 31), "pixel grid too large");
  const int M = (int)M64;
  hipStream_t st = (hipStream_t)stream;
  // dense 3x3 / stride-1 convs whose pixel tiles are whole rows / images: LDS-resident halo tile
  nbdt::HaloGeom hg;
  if (nbdt::conv_halo_applicable(d, M, &hg))
    return nbdt::conv3x3_halo(d, hg, in, w, out, res, bn_scratch, bn, M, st);
  NBDT_REQUIRE(d->wide_tile != 2 && d->wide_tile != 4 && d->wide_tile != 5,
               "wide_tile = 2 / 4 / 5 (force 512-pixel tiles [with the padded LDS pitch] / half tiles): not a dense 3x3 stride-1 conv it fits");
  nbdt::g_last_igemm = "conv_igemm_dma_kernel";
  return nbdt::conv_igemm_dma(d, in, w, out, res, bn_scratch, bn, M, st);
}

static int check_desc(const nbdt_conv_desc* d) {
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
  NBDT_REQUIRE((int64_t)d->B * d->gh * d->gw < (1ll << 31), "pixel grid too large");
  // (ksplit was a reserved, ignored field before version 106: descriptors must be zero-initialised)
  NBDT_REQUIRE(d->ksplit >= 0, "ksplit: 0 (automatic), 1 (never) or n blocks per tile");
  NBDT_REQUIRE(d->ksplit <= 1 || d->wide_tile == 5, "ksplit > 1 goes with wide_tile = 5 (forced half tiles; tests, A/B)");
  return NBDT_OK;
}

extern "C" const char* nbdt_debug_last_igemm(void) { return nbdt::g_last_igemm; }
extern "C" const char* nbdt_debug_last_igemm_full(void) { return nbdt::g_last_igemm_full; }

extern "C" int nbdt_conv_igemm_multi(const nbdt_conv_desc* descs, int32_t n, const void* in, const void* w, void* out,
                                     void* stream) {
  NBDT_REQUIRE(descs && in && w && out, "null argument");
  NBDT_REQUIRE(n >= 1 && n <= 4, "1 to 4 descriptors per launch");
  for (int c = 0; c < n; ++c) {
    int rc = check_desc(descs + c);
    if (rc) return rc;
    NBDT_REQUIRE(descs[c].cout == descs[0].cout && descs[c].cin == descs[0].cin &&
                 descs[c].accumulate == descs[0].accumulate && descs[c].w_ntaps == descs[0].w_ntaps,
                 "the descriptors of one launch share channels, weight layout and accumulate");
  }
  nbdt::g_last_igemm = "conv_igemm_dma_multi_kernel";
  return nbdt::conv_igemm_dma_multi(descs, n, in, w, out, (hipStream_t)stream);
}

extern "C" int nbdt_conv_plan(const nbdt_conv_desc* d, int32_t* form, int32_t* ksplit) {
  NBDT_REQUIRE(d && form && ksplit, "null argument");
  int rc = check_desc(d);
  if (rc) return rc;
  const int64_t M64 = (int64_t)d->B * d->gh * d->gw;
  NBDT_REQUIRE(M64 < (1ll << 31), "pixel grid too large");
  *form = 0;
  *ksplit = 1;
  nbdt::HaloGeom hg;
  if (!nbdt::conv_halo_applicable(d, (int)M64, &hg)) {
    NBDT_REQUIRE(d->wide_tile != 2 && d->wide_tile != 4 && d->wide_tile != 5,
                 "wide_tile = 2 / 4 / 5 (force 512-pixel tiles [with the padded LDS pitch] / half tiles): not a dense 3x3 stride-1 conv it fits");
    return NBDT_OK;
  }
  if (hg.nwv != 8) { *form = 1; return NBDT_OK; }
  if (hg.mw == 1) {
    int nt = 0;
    const int items = nbdt::conv_halo_items(*d, hg, (int)M64, &nt);     // the launcher's own arithmetic
    *form = 4;
    *ksplit = nbdt::conv_ksplit_rule(*d, nt, items);
    return NBDT_OK;
  }
  *form = hg.pad ? 3 : 2;
  return NBDT_OK;
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
