// VERIFICATION-ONLY fp32-storage kernels (gfx950).  Not the product path and never selected by it.
//
// The engines store activations and activation gradients in bf16, and that storage -- not any kernel -- is what
// separates a WRN-28-10 step from the fp32 reference: a 1-ulp bf16 difference flips ~0.3 % of the ReLU masks per
// layer, so whole-step gradients agree with an fp32 oracle only to a cosine of ~0.9.  That
// leaves a question the per-kernel parity tests cannot answer: is the ENGINE's orchestration -- launch order, the
// two-stream schedule with its rotating gradient buffers, which tensor feeds which op, the fused statistics /
// CU-confined BatchNorm backward protocol -- exactly the reference's computation?  engine.set_reference_fp32(True)
// answers it: the same Python path (same forward(), backward(), streams, events, buffers), but every buffer is fp32
// and nbdt.ops routes each launch on an fp32 tensor to the kernel of the same meaning below.  The whole step then has
// to agree with the fp32 oracle to ~1e-5 (tests/test_reference_fp32_gpu.py asserts 1e-3 per parameter gradient).
//
// The kernels are written to be obviously right, not fast: one thread per output element (or per channel), plain
// loops, double-precision accumulators for the long reductions.  They take the SAME descriptors as the product
// kernels (nbdt_conv_desc / nbdt_wgrad_desc tap tables, padded NHWC geometry), so a wrong descriptor shows up here too.
#include "common.h"

using namespace nbdt;

__device__ __forceinline__ int ref_pix(int m, int gh, int gw, int bs, int hs, int ws, int base) {
  const int j = m % gw, t = m / gw;
  return (t / gh) * bs + (t % gh) * hs + j * ws + base;
}

// out[pix_out(m)][n] (+)= sum_t sum_c in[pix_in(m) + tap_off[t] + c] * w[n][w_tap[t]][c]  (+ residual)
__global__ __launch_bounds__(256) void ref_conv_kernel(nbdt_conv_desc d, const float* __restrict__ in,
                                                       const float* __restrict__ w, float* __restrict__ out,
                                                       const float* __restrict__ res, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx % d.cout), m = (int)(idx / d.cout);
  const int pi = ref_pix(m, d.gh, d.gw, d.in_bs, d.in_hs, d.in_ws, d.in_base);
  const int po = ref_pix(m, d.gh, d.gw, d.out_bs, d.out_hs, d.out_ws, d.out_base) + n;
  float acc = 0.f;
  for (int t = 0; t < d.ntaps; ++t) {
    const float* a = in + pi + d.tap_off[t];
    const float* b = w + ((size_t)n * d.w_ntaps + d.w_tap[t]) * d.cin;
    float s = 0.f;
    for (int c = 0; c < d.cin; ++c) s += a[c] * b[c];
    acc += s;
  }
  if (d.accumulate) acc += out[po];
  if (res) acc += res[po];
  out[po] = acc;
}

// dw[co][w_tap[t]][ci] += sum_m gy[pix_g(m)][co] * x[pix_x(m) + tap_off[t] + ci]
__global__ __launch_bounds__(256) void ref_wgrad_kernel(nbdt_wgrad_desc d, const float* __restrict__ x,
                                                        const float* __restrict__ gy, float* __restrict__ dw,
                                                        long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ci = (int)(idx % d.cin);
  const int t = (int)((idx / d.cin) % d.ntaps);
  const int co = (int)(idx / ((long long)d.cin * d.ntaps));
  const int M = d.B * d.gh * d.gw;
  double s = 0.0;
  for (int m = 0; m < M; ++m) {
    const int px = ref_pix(m, d.gh, d.gw, d.x_bs, d.x_hs, d.x_ws, d.x_base);
    const int pg = ref_pix(m, d.gh, d.gw, d.g_bs, d.g_hs, d.g_ws, d.g_base);
    s += (double)gy[pg + co] * (double)x[px + d.tap_off[t] + ci];
  }
  dw[((size_t)co * d.w_ntaps + d.w_tap[t]) * d.cin + ci] += (float)s;
}

// one block per channel: sum / sum of squares over the interior pixels (double accumulators)
__global__ __launch_bounds__(256) void ref_channel_sums_kernel(const float* __restrict__ x, PadGeom g,
                                                               double* __restrict__ sums /* [2][C] */) {
  __shared__ double red[2][256];
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int p = threadIdx.x; p < g.npix; p += 256) {
    const double v = x[pad_offset(g, p) + c];
    s += v; q += v * v;
  }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) { red[0][threadIdx.x] += red[0][threadIdx.x + k]; red[1][threadIdx.x] += red[1][threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { sums[c] = red[0][0]; sums[g.C + c] = red[1][0]; }
}

__global__ void ref_bn_finish_stats_kernel(const double* __restrict__ sums, int C, double n, float eps, float momentum,
                                           float* running_mean, float* running_var, float* save_mean, float* save_rstd,
                                           float* partial_row0 /* nullable: [2][C] fp32 sums (conv-epilogue form) */) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (partial_row0) { partial_row0[c] = (float)sums[c]; partial_row0[C + c] = (float)sums[C + c]; return; }
  const double mean = sums[c] / n;
  double var = sums[C + c] / n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  save_mean[c] = (float)mean;
  save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(256) void ref_bn_apply_kernel(const float* __restrict__ x, const float* mean,
                                                           const float* rstd, const float* gamma, const float* beta,
                                                           const float* __restrict__ res, int relu, PadGeom g,
                                                           float* __restrict__ y) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.npix * g.C) return;
  const int c = (int)(idx % g.C), p = (int)(idx / g.C);
  const int o = pad_offset(g, p) + c;
  const float sc = gamma[c] * rstd[c];
  float v = x[o] * sc + (beta[c] - mean[c] * sc);          // the product kernels' expression (mask recomputation agrees)
  if (res) v += res[o];
  if (relu) v = v > 0.f ? v : 0.f;
  y[o] = v;
}

// masked gradient of one element: relu mask from y when given, else recomputed from x with bn_apply's expression
__device__ __forceinline__ float ref_masked(float gyv, const float* y, int o, float xv, float sc, float sh, int relu) {
  if (!relu) return gyv;
  if (y) return y[o] > 0.f ? gyv : 0.f;
  return (xv * sc + sh) > 0.f ? gyv : 0.f;
}

// one block per channel: sum g', sum g' * xhat  (gy == NULL: gy = gpooled[b][c] / (H*W), the pooled head)
__global__ __launch_bounds__(256) void ref_bn_bwd_sums_kernel(const float* __restrict__ gy, const float* gpooled,
                                                              const float* __restrict__ y, const float* __restrict__ x,
                                                              const float* mean, const float* rstd, const float* gamma,
                                                              const float* beta, int relu, PadGeom g,
                                                              float* dsum, float* dgamma, float* dbeta) {
  __shared__ double red[2][256];
  const int c = blockIdx.x;
  const float mu = mean[c], rs = rstd[c];
  const float sc = gamma[c] * rs, sh = beta ? beta[c] - mu * sc : 0.f;
  const int hw = g.H * g.W;
  double s0 = 0.0, s1 = 0.0;
  for (int p = threadIdx.x; p < g.npix; p += 256) {
    const int o = pad_offset(g, p) + c;
    const float xv = x[o];
    const float gv = gy ? gy[o] : gpooled[(size_t)(p / hw) * g.C + c] / (float)hw;
    const float gg = ref_masked(gv, y, o, xv, sc, sh, relu);
    s0 += gg; s1 += (double)gg * (double)((xv - mu) * rs);
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) { red[0][threadIdx.x] += red[0][threadIdx.x + k]; red[1][threadIdx.x] += red[1][threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dsum[c] = (float)red[0][0]; dsum[g.C + c] = (float)red[1][0];
    if (dbeta) dbeta[c] += (float)red[0][0];
    if (dgamma) dgamma[c] += (float)red[1][0];
  }
}

__global__ __launch_bounds__(256) void ref_bn_bwd_apply_kernel(const float* __restrict__ gy, const float* gpooled,
                                                               const float* __restrict__ y, const float* __restrict__ x,
                                                               const float* mean, const float* rstd, const float* gamma,
                                                               const float* beta, const float* dsum,
                                                               const float* __restrict__ gx_add, int relu, PadGeom g,
                                                               float* __restrict__ gx, float* __restrict__ g_resid) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.npix * g.C) return;
  const int c = (int)(idx % g.C), p = (int)(idx / g.C);
  const int o = pad_offset(g, p) + c;
  const float mu = mean[c], rs = rstd[c];
  const float sc = gamma[c] * rs, sh = beta ? beta[c] - mu * sc : 0.f;
  const float inv_n = 1.f / (float)g.npix;
  const int hw = g.H * g.W;
  const float xv = x[o];
  const float gv = gy ? gy[o] : gpooled[(size_t)(p / hw) * g.C + c] / (float)hw;
  const float gg = ref_masked(gv, y, o, xv, sc, sh, relu);
  float v = sc * (gg - dsum[c] * inv_n - (xv - mu) * rs * dsum[g.C + c] * inv_n);
  if (gx_add) v += gx_add[o];
  gx[o] = v;
  if (g_resid) g_resid[o] = gg;
}

__global__ __launch_bounds__(256) void ref_bn_relu_pool_kernel(const float* __restrict__ x, const float* mean,
                                                               const float* rstd, const float* gamma, const float* beta,
                                                               PadGeom g, float* __restrict__ pooled) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= g.B * g.C) return;
  const int c = idx % g.C, b = idx / g.C;
  const float sc = gamma[c] * rstd[c], sh = beta[c] - mean[c] * sc;
  double s = 0.0;
  for (int p = 0; p < g.H * g.W; ++p) {
    const float v = x[pad_offset(g, b * g.H * g.W + p) + c] * sc + sh;
    s += v > 0.f ? v : 0.f;
  }
  pooled[(size_t)b * g.C + c] = (float)(s / (g.H * g.W));
}

// stem Conv2d(3 -> cout_real, 3x3, pad 1): NCHW fp32 image -> padded NHWC fp32; w [cout][3][3][3] (co, r, s, ci)
__global__ __launch_bounds__(256) void ref_stem_conv_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                            int B, int H, int W, int cout, int cpad, int stride,
                                                            float* __restrict__ out) {
  const int Ho = H / stride, Wo = W / stride;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)B * Ho * Wo * cout) return;
  const int co = (int)(idx % cout);
  const int p = (int)(idx / cout);
  const int xo = p % Wo, yo = (p / Wo) % Ho, b = p / (Wo * Ho);
  float acc = 0.f;
  for (int r = 0; r < 3; ++r)
    for (int s = 0; s < 3; ++s) {
      const int yy = yo * stride + r - 1, xx = xo * stride + s - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      for (int ci = 0; ci < 3; ++ci)
        acc += img[(((size_t)b * 3 + ci) * H + yy) * W + xx] * w[((co * 3 + r) * 3 + s) * 3 + ci];
    }
  out[(((size_t)b * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1) * cpad + co] = acc;
}

__global__ __launch_bounds__(256) void ref_stem_wgrad_kernel(const float* __restrict__ img, const float* __restrict__ gy,
                                                             int B, int H, int W, int cout, int cpad, int stride,
                                                             float* __restrict__ dw) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= cout * 27) return;
  const int ci = idx % 3, s = (idx / 3) % 3, r = (idx / 9) % 3, co = idx / 27;
  const int Ho = H / stride, Wo = W / stride;
  double acc = 0.0;
  for (int b = 0; b < B; ++b)
    for (int yo = 0; yo < Ho; ++yo)
      for (int xo = 0; xo < Wo; ++xo) {
        const int yy = yo * stride + r - 1, xx = xo * stride + s - 1;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        acc += (double)gy[(((size_t)b * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1) * cpad + co] *
               (double)img[(((size_t)b * 3 + ci) * H + yy) * W + xx];
      }
  dw[idx] += (float)acc;
}

// ------------------------------------------------------------------------------------------ host
static double* ref_sums_ws(hipStream_t st, int C) {
  // 2*C doubles per call, from the deterministic-mode workspace (stream-ordered, library-owned)
  return reinterpret_cast<double*>(det_rows(st, (size_t)4 * C + 4));
}

extern "C" int nbdt_ref_conv(const nbdt_conv_desc* d, const float* in, const float* w, float* out,
                             const float* residual, void* stream) {
  NBDT_REQUIRE(d && in && w && out, "null argument");
  const long long total = (long long)d->B * d->gh * d->gw * d->cout;
  hipLaunchKernelGGL(ref_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d, in, w,
                     out, residual, total);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_ref_wgrad(const nbdt_wgrad_desc* d, const float* x, const float* gy, float* dw, void* stream) {
  NBDT_REQUIRE(d && x && gy && dw, "null argument");
  const long long total = (long long)d->cout * d->ntaps * d->cin;
  hipLaunchKernelGGL(ref_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d, x, gy,
                     dw, total);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

/* statistics of a padded fp32 tensor.  partials == NULL: nbdt_bn_stats (mean / rstd / running statistics);
 * partials != NULL: the conv-epilogue form -- row 0 of [rows][2][C] receives the sums, the other rows zero, and
 * nbdt_bn_finalize folds them like the product path's per-tile rows. */
extern "C" int nbdt_ref_bn_stats(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float eps, float momentum,
                                 float* running_mean, float* running_var, float* save_mean, float* save_rstd,
                                 float* partials, void* stream) {
  NBDT_REQUIRE(x && (partials || (save_mean && save_rstd)), "null argument");
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  double* sums = ref_sums_ws(st, C);
  if (!sums) return nbdt::fail(NBDT_ENOMEM, "fp32 reference path: %s (%s)", "no workspace", nbdt::det_rows_why());
  hipLaunchKernelGGL(ref_channel_sums_kernel, dim3(C), dim3(256), 0, st, x, g, sums);
  NBDT_LAUNCH_CHECK();
  if (partials) {
    const size_t rows = ((size_t)g.npix + 255) / 256;
    NBDT_HIP_CHECK(hipMemsetAsync(partials, 0, rows * 2 * C * sizeof(float), st));
  }
  hipLaunchKernelGGL(ref_bn_finish_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, (double)g.npix, eps,
                     momentum, running_mean, running_var, save_mean, save_rstd, partials);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_ref_bn_apply(const float* x, const float* save_mean, const float* save_rstd, const float* gamma,
                                 const float* beta, const float* residual, int32_t relu, int32_t B, int32_t H, int32_t W,
                                 int32_t C, float* y, void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && y, "null argument");
  const PadGeom g = make_geom(B, H, W, C);
  const long long total = (long long)g.npix * C;
  hipLaunchKernelGGL(ref_bn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     save_mean, save_rstd, gamma, beta, residual, relu, g, y);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

// twin of nbdt_bn_apply_s2d: y is the space-to-depth copy [B][H/2+2][W/2+2][4C]
__global__ __launch_bounds__(256) void ref_bn_apply_s2d_kernel(const float* __restrict__ x, const float* mean,
                                                               const float* rstd, const float* gamma, const float* beta,
                                                               int relu, PadGeom g, float* __restrict__ y) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.npix * g.C) return;
  const int c = (int)(idx % g.C), p = (int)(idx / g.C);
  const int w = p % g.W, h = (p / g.W) % g.H, b = p / (g.W * g.H);
  const int o = pad_offset(g, p) + c;
  const long long row2 = (long long)(g.W / 2 + 2) * 4 * g.C, img2 = (g.H / 2 + 2) * row2;
  const long long o2 = b * img2 + ((h >> 1) + 1) * row2 + ((w >> 1) + 1) * 4 * g.C + ((h & 1) * 2 + (w & 1)) * g.C + c;
  const float sc = gamma[c] * rstd[c];
  float v = x[o] * sc + (beta[c] - mean[c] * sc);
  if (relu) v = v > 0.f ? v : 0.f;
  y[o2] = v;
}
extern "C" int nbdt_ref_bn_apply_s2d(const float* x, const float* save_mean, const float* save_rstd, const float* gamma,
                                     const float* beta, int32_t relu, int32_t B, int32_t H, int32_t W, int32_t C,
                                     float* y, void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && y, "null argument");
  NBDT_REQUIRE(H % 2 == 0 && W % 2 == 0, "space-to-depth needs even H and W");
  const PadGeom g = make_geom(B, H, W, C);
  const long long total = (long long)g.npix * C;
  hipLaunchKernelGGL(ref_bn_apply_s2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x, save_mean, save_rstd, gamma, beta, relu, g, y);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

/* whole BatchNorm(+ReLU) backward: sums (dsum, += dgamma, dbeta; reduce != 0) and the elementwise pass.  gy == NULL: the pooled head
 * (gy = gpooled[b][c] / (H*W)); y == NULL with relu: mask recomputed from x (needs beta). */
extern "C" int nbdt_ref_bn_bwd(const float* gy, const float* gpooled, const float* y, const float* x,
                               const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                               int32_t relu, const float* gx_add, int32_t B, int32_t H, int32_t W, int32_t C,
                               int32_t reduce, float* dsum, float* dgamma, float* dbeta, float* gx, float* g_resid,
                               void* stream) {
  NBDT_REQUIRE((gy || gpooled) && x && save_mean && save_rstd && gamma && dsum && gx, "null argument");
  NBDT_REQUIRE(!relu || y || beta, "relu backward needs y, or beta to recompute the mask");
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  if (reduce) {      // reduce == 0: apply only, with the caller's dsum (nbdt_[pool_]bn_bwd_apply)
    hipLaunchKernelGGL(ref_bn_bwd_sums_kernel, dim3(C), dim3(256), 0, st, gy, gpooled, y, x, save_mean, save_rstd, gamma,
                       beta, relu, g, dsum, dgamma, dbeta);
    NBDT_LAUNCH_CHECK();
  }
  const long long total = (long long)g.npix * C;
  hipLaunchKernelGGL(ref_bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gy, gpooled, y, x,
                     save_mean, save_rstd, gamma, beta, dsum, gx_add, relu, g, gx, g_resid);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_ref_bn_relu_pool(const float* x, const float* save_mean, const float* save_rstd, const float* gamma,
                                     const float* beta, int32_t B, int32_t H, int32_t W, int32_t C, float* pooled,
                                     void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && pooled, "null argument");
  const PadGeom g = make_geom(B, H, W, C);
  hipLaunchKernelGGL(ref_bn_relu_pool_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, save_mean,
                     save_rstd, gamma, beta, g, pooled);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_ref_stem_conv(const float* img, const float* w, int32_t B, int32_t H, int32_t W, int32_t cout_real,
                                  int32_t cpad, int32_t stride, float* out, void* stream) {
  NBDT_REQUIRE(img && w && out && (stride == 1 || stride == 2), "bad argument");
  const long long total = (long long)B * (H / stride) * (W / stride) * cout_real;
  hipLaunchKernelGGL(ref_stem_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, w,
                     B, H, W, cout_real, cpad, stride, out);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_ref_stem_wgrad(const float* img, const float* gy, int32_t B, int32_t H, int32_t W, int32_t cout_real,
                                   int32_t cpad, int32_t stride, float* dw, void* stream) {
  NBDT_REQUIRE(img && gy && dw && (stride == 1 || stride == 2), "bad argument");
  hipLaunchKernelGGL(ref_stem_wgrad_kernel, dim3((cout_real * 27 + 255) / 256), dim3(256), 0, (hipStream_t)stream, img, gy,
                     B, H, W, cout_real, cpad, stride, dw);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

// ------------------------------------------------------------------------------------------------------------
// MBConv pieces (EfficientNet-B0, csrc/effnet.hip) in fp32 storage: BatchNorm + activation (+ SE gate, + skip), the
// pooled sums, their backward, and the depthwise convolution with its two gradients.  Same arguments and meaning as the
// product entry points (include/nbdt_hip.h "MBConv pieces"); the squeeze-and-excitation gate, dropout and the linear
// head work on fp32 vectors in the product path already and have no twin here.
__device__ __forceinline__ float ref_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ float ref_act(float v, int act) {
  if (act == NBDT_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == NBDT_ACT_SWISH) return v * ref_sigmoid(v);
  return v;
}
__device__ __forceinline__ float ref_act_grad(float v, int act) {
  if (act == NBDT_ACT_RELU) return v > 0.f ? 1.f : 0.f;
  if (act == NBDT_ACT_SWISH) { const float s = ref_sigmoid(v); return s * (1.f + v * (1.f - s)); }
  return 1.f;
}

// y = act(bn(x)) [* gate[b][c]] [+ residual]
__global__ __launch_bounds__(256) void ref_bn_act_apply_kernel(const float* __restrict__ x, const float* mean,
                                                               const float* rstd, const float* gamma, const float* beta,
                                                               int act, const float* __restrict__ gate,
                                                               const float* __restrict__ res, PadGeom g,
                                                               float* __restrict__ y) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.npix * g.C) return;
  const int c = (int)(idx % g.C), p = (int)(idx / g.C);
  const int o = pad_offset(g, p) + c, b = p / (g.H * g.W);
  const float sc = gamma[c] * rstd[c];
  float v = ref_act(x[o] * sc + (beta[c] - mean[c] * sc), act);
  if (gate) v *= gate[(size_t)b * g.C + c];
  if (res) v += res[o];
  y[o] = v;
}

// out[b][c] = scale * sum_hw act(bn(x)) [* mul]
__global__ __launch_bounds__(256) void ref_bn_act_pool_kernel(const float* __restrict__ x, const float* mean,
                                                              const float* rstd, const float* gamma, const float* beta,
                                                              int act, const float* __restrict__ mul, float scale,
                                                              PadGeom g, float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= g.B * g.C) return;
  const int c = idx % g.C, b = idx / g.C, hw = g.H * g.W;
  const float sc = gamma[c] * rstd[c], sh = beta[c] - mean[c] * sc;
  double s = 0.0;
  for (int p = 0; p < hw; ++p) {
    const int o = pad_offset(g, b * hw + p) + c;
    float v = ref_act(x[o] * sc + sh, act);
    if (mul) v *= mul[o];
    s += v;
  }
  out[(size_t)b * g.C + c] = (float)(s * (double)scale);
}

// gradient entering the activation of element o of image b, channel c (include/nbdt_hip.h, nbdt_bn_act_bwd)
__device__ __forceinline__ float ref_act_in_grad(const float* gu, const float* gate, const float* gpool, int o, int b,
                                                 int c, int C, int hw) {
  if (!gu) return gpool[(size_t)b * C + c] / (float)hw;
  float ga = gu[o];
  if (gate) ga = ga * gate[(size_t)b * C + c] + gpool[(size_t)b * C + c] / (float)hw;
  return ga;
}

__global__ __launch_bounds__(256) void ref_bn_act_bwd_sums_kernel(const float* __restrict__ gu, const float* gate,
                                                                  const float* gpool, const float* __restrict__ x,
                                                                  const float* mean, const float* rstd,
                                                                  const float* gamma, const float* beta, int act,
                                                                  PadGeom g, float* dsum, float* dgamma, float* dbeta) {
  __shared__ double red[2][256];
  const int c = blockIdx.x, hw = g.H * g.W;
  const float mu = mean[c], rs = rstd[c];
  const float sc = gamma[c] * rs, sh = beta[c] - mu * sc;
  double s0 = 0.0, s1 = 0.0;
  for (int p = threadIdx.x; p < g.npix; p += 256) {
    const int o = pad_offset(g, p) + c;
    const float xv = x[o];
    const float gg = ref_act_in_grad(gu, gate, gpool, o, p / hw, c, g.C, hw) * ref_act_grad(xv * sc + sh, act);
    s0 += gg; s1 += (double)gg * (double)((xv - mu) * rs);
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) { red[0][threadIdx.x] += red[0][threadIdx.x + k]; red[1][threadIdx.x] += red[1][threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dsum[c] = (float)red[0][0]; dsum[g.C + c] = (float)red[1][0];
    if (dbeta) dbeta[c] += (float)red[0][0];
    if (dgamma) dgamma[c] += (float)red[1][0];
  }
}

// (gx may be gu: every element is read and written by the one thread that owns it)
__global__ __launch_bounds__(256) void ref_bn_act_bwd_apply_kernel(const float* gu, const float* gate, const float* gpool,
                                                                   const float* __restrict__ x, const float* mean,
                                                                   const float* rstd, const float* gamma,
                                                                   const float* beta, int act, const float* dsum,
                                                                   const float* gx_add, PadGeom g, float* gx) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.npix * g.C) return;
  const int c = (int)(idx % g.C), p = (int)(idx / g.C), hw = g.H * g.W;
  const int o = pad_offset(g, p) + c;
  const float mu = mean[c], rs = rstd[c];
  const float sc = gamma[c] * rs, sh = beta[c] - mu * sc;
  const float inv_n = 1.f / (float)g.npix;
  const float xv = x[o];
  const float gg = ref_act_in_grad(gu, gate, gpool, o, p / hw, c, g.C, hw) * ref_act_grad(xv * sc + sh, act);
  float v = sc * (gg - dsum[c] * inv_n - (xv - mu) * rs * dsum[g.C + c] * inv_n);
  if (gx_add) v += gx_add[o];
  gx[o] = v;
}

// depthwise Conv2d(C, C, k, stride, padding k/2, groups = C); w [k*k][C]; x [B][H+2][W+2][C] -> y [B][H/s+2][W/s+2][C]
__global__ __launch_bounds__(256) void ref_dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, int B,
                                                         int H, int W, int C, int k, int stride, float* __restrict__ y) {
  const int Ho = H / stride, Wo = W / stride, pad = k / 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)B * Ho * Wo * C) return;
  const int c = (int)(idx % C);
  const int p = (int)(idx / C);
  const int wo = p % Wo, ho = (p / Wo) % Ho, b = p / (Wo * Ho);
  float acc = 0.f;
  for (int r = 0; r < k; ++r)
    for (int s = 0; s < k; ++s) {
      const int hi = ho * stride + r - pad, wi = wo * stride + s - pad;
      if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
      acc += x[(((size_t)b * (H + 2) + hi + 1) * (W + 2) + wi + 1) * C + c] * w[(size_t)(r * k + s) * C + c];
    }
  y[(((size_t)b * (Ho + 2) + ho + 1) * (Wo + 2) + wo + 1) * C + c] = acc;
}

// gx[hi][wi] = sum over (r, s) with (hi + pad - r, wi + pad - s) = stride * (ho, wo) of gy[ho][wo] * w[r][s]
__global__ __launch_bounds__(256) void ref_dw_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                              int B, int H, int W, int C, int k, int stride,
                                                              float* __restrict__ gx) {
  const int Ho = H / stride, Wo = W / stride, pad = k / 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)B * H * W * C) return;
  const int c = (int)(idx % C);
  const int p = (int)(idx / C);
  const int wi = p % W, hi = (p / W) % H, b = p / (W * H);
  float acc = 0.f;
  for (int r = 0; r < k; ++r) {
    const int th = hi + pad - r;
    if (th < 0 || th % stride != 0 || th / stride >= Ho) continue;
    for (int s = 0; s < k; ++s) {
      const int tw = wi + pad - s;
      if (tw < 0 || tw % stride != 0 || tw / stride >= Wo) continue;
      acc += gy[(((size_t)b * (Ho + 2) + th / stride + 1) * (Wo + 2) + tw / stride + 1) * C + c] *
             w[(size_t)(r * k + s) * C + c];
    }
  }
  gx[(((size_t)b * (H + 2) + hi + 1) * (W + 2) + wi + 1) * C + c] = acc;
}

// dw[r*k + s][c] += sum over images and output pixels of gy[ho][wo] * x[ho*stride + r - pad][wo*stride + s - pad]
__global__ __launch_bounds__(256) void ref_dw_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                int B, int H, int W, int C, int k, int stride,
                                                                float* __restrict__ dw) {
  const int Ho = H / stride, Wo = W / stride, pad = k / 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= k * k * C) return;
  const int c = idx % C, t = idx / C;
  const int r = t / k, s = t % k;
  double acc = 0.0;
  for (int b = 0; b < B; ++b)
    for (int ho = 0; ho < Ho; ++ho) {
      const int hi = ho * stride + r - pad;
      if (hi < 0 || hi >= H) continue;
      for (int wo = 0; wo < Wo; ++wo) {
        const int wi = wo * stride + s - pad;
        if (wi < 0 || wi >= W) continue;
        acc += (double)gy[(((size_t)b * (Ho + 2) + ho + 1) * (Wo + 2) + wo + 1) * C + c] *
               (double)x[(((size_t)b * (H + 2) + hi + 1) * (W + 2) + wi + 1) * C + c];
      }
    }
  dw[idx] += (float)acc;
}

extern "C" int nbdt_ref_bn_act_apply(const float* x, const float* save_mean, const float* save_rstd, const float* gamma,
                                     const float* beta, int32_t act, const float* gate, const float* residual, int32_t B,
                                     int32_t H, int32_t W, int32_t C, float* y, void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && y, "null argument");
  const PadGeom g = make_geom(B, H, W, C);
  const long long total = (long long)g.npix * C;
  hipLaunchKernelGGL(ref_bn_act_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     save_mean, save_rstd, gamma, beta, act, gate, residual, g, y);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_ref_bn_act_pool(const float* x, const float* save_mean, const float* save_rstd, const float* gamma,
                                    const float* beta, int32_t act, const float* mul, float scale, int32_t B, int32_t H,
                                    int32_t W, int32_t C, float* out, void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && out, "null argument");
  const PadGeom g = make_geom(B, H, W, C);
  hipLaunchKernelGGL(ref_bn_act_pool_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, save_mean,
                     save_rstd, gamma, beta, act, mul, scale, g, out);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

/* nbdt_bn_act_bwd (sums + elementwise pass; reduce == 0: the pass only, with the caller's dsum -- not used by the engine:
 * its fused-sums form recomputes them here, see nbdt.ops.bn_act_bwd_apply) */
extern "C" int nbdt_ref_bn_act_bwd(const float* gu, const float* gate, const float* gpool, const float* x,
                                   const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                                   int32_t act, const float* gx_add, int32_t B, int32_t H, int32_t W, int32_t C,
                                   float* dsum, float* dgamma, float* dbeta, float* gx, void* stream) {
  NBDT_REQUIRE(x && save_mean && save_rstd && gamma && beta && dsum && gx, "null argument");
  NBDT_REQUIRE((gu && !gate && !gpool) || (gu && gate && gpool) || (!gu && !gate && gpool),
               "gradient forms: gu | gu, gate, gpool | gpool");
  hipStream_t st = (hipStream_t)stream;
  const PadGeom g = make_geom(B, H, W, C);
  hipLaunchKernelGGL(ref_bn_act_bwd_sums_kernel, dim3(C), dim3(256), 0, st, gu, gate, gpool, x, save_mean, save_rstd,
                     gamma, beta, act, g, dsum, dgamma, dbeta);
  NBDT_LAUNCH_CHECK();
  const long long total = (long long)g.npix * C;
  hipLaunchKernelGGL(ref_bn_act_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gu, gate, gpool,
                     x, save_mean, save_rstd, gamma, beta, act, dsum, gx_add, g, gx);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

static int ref_check_dw(int B, int H, int W, int C, int k, int stride) {
  NBDT_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2) &&
               H % stride == 0 && W % stride == 0, "bad depthwise geometry");
  return NBDT_OK;
}

extern "C" int nbdt_ref_dwconv_fwd(const float* x, const float* w, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                                   int32_t stride, float* y, void* stream) {
  NBDT_REQUIRE(x && w && y, "null argument");
  if (int rc = ref_check_dw(B, H, W, C, k, stride)) return rc;
  const long long total = (long long)B * (H / stride) * (W / stride) * C;
  hipLaunchKernelGGL(ref_dw_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, B,
                     H, W, C, k, stride, y);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

/* H, W: the INPUT-sized gradient being written (like nbdt_dwconv_bwd_data) */
extern "C" int nbdt_ref_dwconv_bwd_data(const float* gy, const float* w, int32_t B, int32_t H, int32_t W, int32_t C,
                                        int32_t k, int32_t stride, float* gx, void* stream) {
  NBDT_REQUIRE(gy && w && gx, "null argument");
  if (int rc = ref_check_dw(B, H, W, C, k, stride)) return rc;
  const long long total = (long long)B * H * W * C;
  hipLaunchKernelGGL(ref_dw_bwd_data_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gy,
                     w, B, H, W, C, k, stride, gx);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_ref_dwconv_bwd_weight(const float* x, const float* gy, int32_t B, int32_t H, int32_t W, int32_t C,
                                          int32_t k, int32_t stride, float* dw, void* stream) {
  NBDT_REQUIRE(x && gy && dw, "null argument");
  if (int rc = ref_check_dw(B, H, W, C, k, stride)) return rc;
  hipLaunchKernelGGL(ref_dw_bwd_weight_kernel, dim3((k * k * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, gy, B,
                     H, W, C, k, stride, dw);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}
