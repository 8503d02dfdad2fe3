// Stem convolution, classifier head (nn.Linear) and the SGD update on gfx950.
//
//   stem ....... nn.Conv2d(3, 16|64, 3, padding=1) on NCHW fp32 images (reference
//                nbdt/models/resnet.py:120; pytorchcv CIFARWRN init_block).  K = 27 is far too small
//                for MFMA and is 0.03% of the step's flops: direct fp32 FMA kernel that also
//                converts NCHW fp32 -> padded NHWC bf16 on the way out.
//   linear ..... nn.Linear(640|512, classes) forward/backward (resnet.py:126,148), fp32.
//   sgd ........ optim.SGD(momentum=0.9, weight_decay=5e-4) (main.py:207) over ONE flat fp32 buffer
//                holding every parameter; the same pass refreshes the bf16 copy the convs read.
// All HBM/latency-bound; none is on the MFMA critical path.
#include "common.h"

#include <atomic>
#include <map>
#include <utility>

using namespace nbdt;

// ------------------------------------------------------------------------------------------ deterministic mode
namespace nbdt {
static std::atomic<int> g_deterministic{0};
bool deterministic() { return g_deterministic.load(std::memory_order_relaxed) != 0; }

struct DetBuf { float* ptr = nullptr; size_t floats = 0; bool captured = false; };
static std::mutex g_det_mutex;
static std::map<std::pair<int, hipStream_t>, DetBuf> g_det_bufs;

static thread_local char g_det_why[200] = "";
const char* det_rows_why() { return g_det_why; }

float* det_rows(hipStream_t st, size_t floats) {
  int dev = 0;
  snprintf(g_det_why, sizeof(g_det_why), "hipMalloc / hipGetDevice failed");
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_det_mutex);
  DetBuf& b = g_det_bufs[std::make_pair(dev, st)];
  // hipGraph capture: hipMalloc / hipFree / hipStreamSynchronize are illegal while `st` is capturing, and a captured
  // launch bakes this pointer in -- so inside a capture the workspace must already be large enough, and a workspace a
  // capture has used is never freed again (a later, larger request gets a NEW buffer for eager launches only by
  // failing loudly here: the caller releases the graph first).
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  if (b.floats < floats) {
    if (capturing || b.captured) {
      snprintf(g_det_why, sizeof(g_det_why), "the per-stream workspace (%zu floats) cannot grow to %zu %s: run one eager "
               "step at the largest batch size before capturing", b.floats, floats,
               capturing ? "inside a hipGraph capture" : "after a captured graph baked its address in");
      return nullptr;
    }
    // growing: the old buffer may still be read by work queued on this stream
    if (b.ptr) { if (hipStreamSynchronize(st) != hipSuccess) return nullptr; (void)hipFree(b.ptr); b.ptr = nullptr; b.floats = 0; }
    size_t want = floats < (16u << 20) ? (16u << 20) : floats + floats / 4;     // >= 64 MB, then 25 % headroom
    if (hipMalloc((void**)&b.ptr, want * sizeof(float)) != hipSuccess) { b.ptr = nullptr; return nullptr; }
    b.floats = want;
  }
  if (capturing) b.captured = true;
  return b.ptr;
}

__global__ __launch_bounds__(256) void det_fold_kernel(const float* __restrict__ rows, int nrows, size_t n,
                                                       float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = dst[i];
  for (int r = 0; r < nrows; ++r) s += rows[(size_t)r * n + i];      // fixed order: row 0, 1, 2, ...
  dst[i] = s;
}

// four elements per thread (every weight tensor here is a multiple of 4 long and 16-byte aligned)
__global__ __launch_bounds__(256) void det_fold4_kernel(const float4* __restrict__ rows, int nrows, size_t n4,
                                                        float4* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 s = dst[i];
  for (int r = 0; r < nrows; ++r) {
    const float4 v = rows[(size_t)r * n4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  dst[i] = s;
}

// many rows of few elements (the stem weight gradient: ~1000 blocks x 27 * cout sums): 16 row lanes x 16 columns per block,
// every lane sums its rows r = lane, lane + 16, ... in order, then the 16 lanes of a column are added in lane order
__global__ __launch_bounds__(256) void det_fold_tall_kernel(const float* __restrict__ rows, int nrows, size_t n,
                                                            float* __restrict__ dst) {
  __shared__ float part[16][17];
  const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + col;
  float s = 0.f;
  if (i < n)
    for (int r = rl; r < nrows; r += 16) s += rows[(size_t)r * n + i];
  part[rl][col] = s;
  __syncthreads();
  if (rl == 0 && i < n) {
    float t = dst[i];
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][col];
    dst[i] = t;
  }
}

int det_fold(hipStream_t st, const float* rows, int nrows, size_t n, float* dst) {
  if (nrows >= 64 && n <= 65536) {
    hipLaunchKernelGGL(det_fold_tall_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, rows, nrows, n, dst);
    NBDT_LAUNCH_CHECK();
    return NBDT_OK;
  }
  if (n % 4 == 0 && (reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0) {
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(det_fold4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float4*)rows, nrows,
                       n4, (float4*)dst);
    NBDT_LAUNCH_CHECK();
    return NBDT_OK;
  }
  hipLaunchKernelGGL(det_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rows, nrows, n, dst);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}
}  // namespace nbdt

extern "C" int nbdt_set_deterministic(int32_t on) {
  nbdt::g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed);
  return NBDT_OK;
}
extern "C" int nbdt_get_deterministic(void) { return nbdt::deterministic() ? 1 : 0; }

namespace nbdt {
static std::atomic<int> g_wgrad_store{1};      // profiles/r05_wgrad_store_epilogue_ab.txt
bool wgrad_store_epilogue() { return g_wgrad_store.load(std::memory_order_relaxed) != 0; }
}  // namespace nbdt
extern "C" int nbdt_set_wgrad_store_epilogue(int32_t on) {
  nbdt::g_wgrad_store.store(on ? 1 : 0, std::memory_order_relaxed);
  return NBDT_OK;
}
extern "C" int nbdt_get_wgrad_store_epilogue(void) { return nbdt::wgrad_store_epilogue() ? 1 : 0; }

namespace nbdt {
static std::atomic<int> g_reserved_cus{0};
int reserved_cus() { return g_reserved_cus.load(std::memory_order_relaxed); }
}  // namespace nbdt
extern "C" int nbdt_set_reserved_cus(int32_t n) {
  NBDT_REQUIRE(n >= 0 && n <= 128, "reserved CUs must be 0..128");
  nbdt::g_reserved_cus.store(n, std::memory_order_relaxed);
  return NBDT_OK;
}
extern "C" int nbdt_get_reserved_cus(void) { return nbdt::reserved_cus(); }

// ------------------------------------------------------------------------------------------ stem
// thread = one output pixel: its 27 image values stay in registers and every 8-cout chunk re-uses them; the weights sit
// in LDS as [tap][cout], so a chunk's 8 weights of a tap are two broadcast ds_read_b128.  (Round 1's form -- one thread
// per (pixel, 8-cout chunk), 216 ds_read_b32 and 27 global loads each -- took 55 us for the 16-channel WRN stem at 512
// images and 115 us for ResNet18's 64-channel stem at 128 x 64x64.)
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        int B, int H, int W, int cout, int cpad, int stride,
                                                        bf16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [27][cout]
  for (int i = threadIdx.x; i < cout * 27; i += 256) {
    const int co = i / 27, t = i - co * 27;
    wl[t * cout + co] = w[i];
  }
  __syncthreads();
  const int Ho = H / stride, Wo = W / stride;   // H, W: image size; output is Ho x Wo
  const long long total = (long long)B * Ho * Wo;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= total) return;
  const int x = (int)(p % Wo), y = (int)((p / Wo) % Ho), b = (int)(p / ((long long)Wo * Ho));
  float v[27];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int yy = y * stride + r - 1;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int xx = x * stride + s - 1;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
        v[(r * 3 + s) * 3 + ci] = in ? img[(((size_t)b * 3 + ci) * H + yy) * W + xx] : 0.f;
    }
  }
  bf16_t* o = out + (((size_t)b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * cpad;
  for (int ck = 0; ck < cout / 8; ++ck) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const float4 w0 = *(const float4*)(wl + t * cout + ck * 8);
      const float4 w1 = *(const float4*)(wl + t * cout + ck * 8 + 4);
      acc[0] += v[t] * w0.x; acc[1] += v[t] * w0.y; acc[2] += v[t] * w0.z; acc[3] += v[t] * w0.w;
      acc[4] += v[t] * w1.x; acc[5] += v[t] * w1.y; acc[6] += v[t] * w1.z; acc[7] += v[t] * w1.w;
    }
    *(u32x4_t*)(o + ck * 8) = pack8(acc);
  }
}

// dw[co][27] += sum_pixels gy[pix][co] * img[tap], on the matrix pipes (round 5).  Rounds 2-4 had thread (k, co quad)
// accumulate 4 outputs with one ds_read_b32 + one ds_read_b128 per pixel: two LDS words per four multiply-adds, and a CU
// completes one LDS read instruction per ~8 cycles -- 117 us for ResNet18's stem at 128 x 64x64, of which the 67 MB of gy are
// 17.  Here dw^T = sum over pixels of gy[pix][co] (x) patch[pix][k] goes through v_mfma_f32_32x32x2_f32 -- fp32 operands,
// the arithmetic of the scalar form -- with K = pixels: per 64-pixel tile the gy tile [64][cout] and the patch tile [64][27 -> 32]
// go to LDS (a thread owns one pixel of the tile for both loads: one (b, y, x) decomposition per tile), wave w of the block
// owns 16 of the tile's pixels, per pixel PAIR one ds_read_b32 of the patch row and one per 32 couts of the gy row feed
// one MFMA each: 24 reads per wave and tile instead of 256.  The four waves' 32 x 32 x CO_T sums meet in LDS at the end and
// every block writes ONE row of sums; det_fold adds the ~1000 rows to dw in block order (atomics from 1024 blocks onto
// the same 27 * cout addresses were a fixed 43 us of the launch).  WRN stem at 512 images 90 -> 52 us, ResNet18's at
// 128 x 64x64 117 -> 72, EfficientNet-B0's (stride 2, 128 x 224x224) 177 -> 115.
typedef __attribute__((ext_vector_type(16))) float stem_f32x16;
template <int CO_T>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ img, const bf16_t* __restrict__ gy,
                                                              int B, int H, int W, int cout, int cpad, int stride,
                                                              int tiles_per_block, float* __restrict__ dw, int row_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // gy tile [64][gp], patch tile [64][33]; then [4][CO_T][1024]
  // row pitches: a thread owns a PIXEL when it fills the tiles, so lane l writes row l -- pitch 32 would put all 64 lanes of
  // a patch store on two banks (that version ran at the scalar kernel's speed), cout + 32 the float4 stores of gy likewise.
  // cout + 4 and 33: conflict-free stores, and the two pixels of a pair at most 2-way on the 24 reads per wave and tile.
  const int gp = cout + 4;
  constexpr int PP = 33;
  float* gl = lds;
  float* pl = lds + 64 * gp;
  const int Ho = H / stride, Wo = W / stride;
  const int npix = B * Ho * Wo;
  const int c8 = cout / 8;
  const int lp = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l32 = lane & 31, half = lane >> 5;
  stem_f32x16 acc[CO_T];
#pragma unroll
  for (int c = 0; c < CO_T; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int t = 0; t < tiles_per_block; ++t) {
    const int p0 = (blockIdx.x * tiles_per_block + t) * 64;
    if (p0 >= npix) break;
    __syncthreads();
    {
      // every load unconditional on a clamped address, the mask applied to the VALUE: a `live ? load : 0` makes hipcc wait
      // for each load before it issues the next (round 1's lesson) -- 8 + 2 serial HBM round trips per tile
      const int pp_raw = p0 + lp;
      const bool live = pp_raw < npix;
      const int pp = live ? pp_raw : npix - 1;
      const int x = pp % Wo, y = (pp / Wo) % Ho, b = pp / (Wo * Ho);
      constexpr int GMAX = 3;                    // cout <= 72: at most 9 chunks of 8 channels, 4 parts
      u32x4_t gv[GMAX];
#pragma unroll
      for (int i = 0; i < GMAX; ++i) {
        const int ck = part + 4 * i;
        const int ckc = ck < c8 ? ck : c8 - 1;
        gv[i] = *(const u32x4_t*)(gy + (((size_t)b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * cpad + ckc * 8);
      }
      float pv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = part + 4 * i;              // 0..31; columns 27..31 are zero
        const int kc = k < 27 ? k : 26;
        const int r = kc / 9, s2 = (kc / 3) % 3, ci = kc % 3;
        const int yy = y * stride + r - 1, xx = x * stride + s2 - 1;
        const bool in = k < 27 && live && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        const float v = img[(((size_t)b * 3 + ci) * H + yc) * W + xc];
        pv[i] = in ? v : 0.f;
      }
#pragma unroll
      for (int i = 0; i < GMAX; ++i) {
        const int ck = part + 4 * i;
        if (ck < c8) {
          float f[8];
          unpack8(gv[i], f);
          const float m = live ? 1.f : 0.f;
          *(float4*)(gl + lp * gp + ck * 8) = make_float4(f[0] * m, f[1] * m, f[2] * m, f[3] * m);
          *(float4*)(gl + lp * gp + ck * 8 + 4) = make_float4(f[4] * m, f[5] * m, f[6] * m, f[7] * m);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) pl[lp * PP + part + 4 * i] = pv[i];
    }
    __syncthreads();
    // A[i = co][k = pixel of the pair], B[k][j = tap]: lane l holds A[l & 31][l >> 5] and B[l >> 5][l & 31]
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) {
      const int pix = wave * 16 + pr * 2 + half;
      const float bv = pl[pix * PP + l32];
#pragma unroll
      for (int c = 0; c < CO_T; ++c) {
        const int co = c * 32 + l32;
        const float av = co < cout ? gl[pix * gp + co] : 0.f;
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
      }
    }
  }
  // ---- the four waves' sums -> LDS -> one add per (co, k) and block.  acc[c][r]: co = c*32 + 8*(r/4) + 4*half + r%4, k = l32
  __syncthreads();
  float* red = lds + wave * (CO_T * 1024);
#pragma unroll
  for (int c = 0; c < CO_T; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[c * 1024 + (8 * (r / 4) + 4 * half + (r % 4)) * 32 + l32] = acc[c][r];
  __syncthreads();
  for (int o = threadIdx.x; o < CO_T * 1024; o += 256) {
    const int co = o >> 5, k = o & 31;
    if (co < cout && k < 27) {
      const float s4 = (lds[o] + lds[CO_T * 1024 + o]) + (lds[2 * CO_T * 1024 + o] + lds[3 * CO_T * 1024 + o]);
      // row_stride != 0: this block's own row of the workspace, every entry written exactly once (det_fold adds the rows
      // to dw in block order); 0: no workspace, atomics into dw -- ~1000 blocks on the same 27 * cout addresses
      if (row_stride) dw[(size_t)blockIdx.x * row_stride + co * 27 + k] = s4;
      else atomicAdd(dw + co * 27 + k, s4);
    }
  }
}

extern "C" int nbdt_stem_conv(const float* img, const float* w, int32_t B, int32_t H, int32_t W, int32_t cout_real,
                              int32_t cpad, int32_t stride, void* out, void* stream) {
  NBDT_REQUIRE(img && w && out, "null argument");
  NBDT_REQUIRE(B > 0 && H > 0 && W > 0, "empty image batch");
  NBDT_REQUIRE((stride == 1 || stride == 2) && H % stride == 0 && W % stride == 0, "bad stem stride");
  NBDT_REQUIRE(cout_real > 0 && cout_real % 8 == 0 && cout_real <= cpad && cpad % 8 == 0, "bad stem channels");
  const long long total = (long long)B * (H / stride) * (W / stride);      // one thread per output pixel
  hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), cout_real * 27 * sizeof(float),
                     (hipStream_t)stream, img, w, B, H, W, cout_real, cpad, stride, (bf16_t*)out);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_stem_wgrad(const float* img, const void* gy, int32_t B, int32_t H, int32_t W, int32_t cout_real,
                               int32_t cpad, int32_t stride, float* dw, void* stream) {
  NBDT_REQUIRE(img && gy && dw, "null argument");
  NBDT_REQUIRE((stride == 1 || stride == 2) && H % stride == 0 && W % stride == 0, "bad stem stride");
  NBDT_REQUIRE(cout_real > 0 && cout_real % 8 == 0 && cout_real <= 72 && cout_real <= cpad,
               "stem wgrad supports cout = 8, 16, ..., 72");
  const int npix = B * (H / stride) * (W / stride);
  const int tiles = (npix + 63) / 64;
  int blocks = tiles < 1024 ? tiles : 1024;
  const int tpb = (tiles + blocks - 1) / blocks;
  blocks = (tiles + tpb - 1) / tpb;
  const int co_t = (cout_real + 31) / 32;
  size_t shmem = (size_t)(64 * (cout_real + 4) + 64 * 33) * sizeof(float);
  if (shmem < (size_t)4 * co_t * 1024 * sizeof(float)) shmem = (size_t)4 * co_t * 1024 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const int nout = cout_real * 27;
  // one row of sums per block + a fold in block order (run-to-run identical bits); atomics only without a workspace
  // (first use inside a hipGraph capture) -- and never in deterministic mode
  float* target = det_rows(st, (size_t)blocks * nout);
  if (!target) {
    if (deterministic()) return nbdt::fail(NBDT_ENOMEM, "deterministic mode: %s (%s)", "no workspace for the per-block rows", nbdt::det_rows_why());
    target = dw;
  }
#define NBDT_STEM(T)                                                                                                  \
  hipLaunchKernelGGL(stem_wgrad_mfma_kernel<T>, dim3(blocks), dim3(256), shmem, st, img, (const bf16_t*)gy, B, H, W,  \
                     cout_real, cpad, stride, tpb, target, target == dw ? 0 : nout)
  if (co_t == 1) NBDT_STEM(1); else if (co_t == 2) NBDT_STEM(2); else NBDT_STEM(3);
#undef NBDT_STEM
  NBDT_LAUNCH_CHECK();
  if (target != dw) return det_fold(st, target, blocks, (size_t)nout, dw);
  return NBDT_OK;
}

// ------------------------------------------------------------------------------------------ linear
// one wave per output (b, n): lanes split K, shuffle reduce; rows of x and w are read coalesced
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int B, int K, int N,
                                                         float* __restrict__ z) {
  const long long o = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= (long long)B * N) return;
  const int b = (int)(o / N), n = (int)(o % N);
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += x[(size_t)b * K + k] * w[(size_t)n * K + k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) z[o] = s + (bias ? bias[n] : 0.f);
}

// gx[b][k] = sum_n gz[b][n] w[n][k]
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* __restrict__ gz, const float* __restrict__ w,
                                                           int B, int K, int N, float* __restrict__ gx) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)B * K) return;
  const int b = (int)(idx / K), k = (int)(idx % K);
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += gz[(size_t)b * N + n] * w[(size_t)n * K + k];
  gx[idx] = s;
}

// gw[n][k] += sum_b gz[b][n] x[b][k]; gb[n] += sum_b gz[b][n]; batch split over grid.y with atomics
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* __restrict__ gz, const float* __restrict__ x,
                                                           int B, int K, int N, int b_per_block,
                                                           float* __restrict__ gw, float* __restrict__ gb) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)N * (K + 1)) return;
  const int n = (int)(idx / (K + 1)), k = (int)(idx % (K + 1));
  const int b0 = blockIdx.y * b_per_block;
  const int b1 = b0 + b_per_block < B ? b0 + b_per_block : B;
  float s = 0.f;
  if (k < K) {
    for (int b = b0; b < b1; ++b) s += gz[(size_t)b * N + n] * x[(size_t)b * K + k];
    atomicAdd(gw + (size_t)n * K + k, s);
  } else if (gb) {
    for (int b = b0; b < b1; ++b) s += gz[(size_t)b * N + n];
    atomicAdd(gb + n, s);
  }
}

// Wide heads (EfficientNet: 1280 -> 1000) : LDS-tiled fp32 GEMM, 64x64 outputs per block, 4x4 per thread.
//   C[i][j] (+)= sum_l A(i,l) * B(l,j) [+ bias[j]];   A(i,l) = A[i*ai + l*al],  B(l,j) = B[l*bl + j*bj]
// (the one-wave-per-output kernels above took 105 / 204 / 133 us for fwd / dgrad / wgrad at B=128)
// At batch 128 the forward / data-gradient grids are 32 / 40 blocks, one wave per SIMD: the K loop's global loads
// are what such a block waits for.  Steps of 32 with the NEXT step's 16 elements per thread already in registers
// while this step's products are formed (round 4: 186 -> see profiles/r04_head_ab.txt); every output is still one
// chain of multiply-adds in ascending l, so the results are the bits the 16-step kernel gave.
// T: tile edge (64: 4x4 outputs per thread; 32: 2x2 -- four times the blocks when a 64-tile grid would leave most CUs
// idle), BK: K elements per step.
template <bool ACC, int T, int BK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long long ai, long long al,
                                                       const float* __restrict__ Bm, long long bl, long long bj,
                                                       const float* __restrict__ bias, float* __restrict__ C,
                                                       int ldc, int M, int N, int L) {
  constexpr int Q = BK * T / 256;                  // elements per thread per operand tile
  constexpr int R = T / 16;                        // thread owns R x R outputs: rows ti*R.., cols tj*R..
  __shared__ float As[BK][T + 4];
  __shared__ float Bs[BK][T + 4];
  const int tid = threadIdx.x;
  const int i0 = blockIdx.y * T, j0 = blockIdx.x * T;
  const int ti = tid >> 4, tj = tid & 15;
  float acc[R][R];
#pragma unroll
  for (int a = 0; a < R; ++a)
#pragma unroll
    for (int b = 0; b < R; ++b) acc[a][b] = 0.f;
  float ra[Q], rb[Q];
  // element e of a tile <-> (row / col, l): consecutive threads walk the operand's unit-stride dimension
  auto a_pos = [&](int e, int& ia, int& la) { if (al == 1) { la = e & (BK - 1); ia = e / BK; } else { ia = e & (T - 1); la = e / T; } };
  auto b_pos = [&](int e, int& jb, int& lb) { if (bj == 1) { jb = e & (T - 1); lb = e / T; } else { lb = e & (BK - 1); jb = e / BK; } };
  auto fetch = [&](int l0) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      int ia, la, jb, lb;
      a_pos(tid + 256 * q, ia, la);
      b_pos(tid + 256 * q, jb, lb);
      const int gi = i0 + ia, gl = l0 + la, gj = j0 + jb, gl2 = l0 + lb;
      ra[q] = (gi < M && gl < L) ? A[gi * ai + gl * al] : 0.f;
      rb[q] = (gj < N && gl2 < L) ? Bm[gl2 * bl + gj * bj] : 0.f;
    }
  };
  fetch(0);
  for (int l0 = 0; l0 < L; l0 += BK) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      int ia, la, jb, lb;
      a_pos(tid + 256 * q, ia, la);
      b_pos(tid + 256 * q, jb, lb);
      As[la][ia] = ra[q];
      Bs[lb][jb] = rb[q];
    }
    __syncthreads();
    if (l0 + BK < L) fetch(l0 + BK);
#pragma unroll
    for (int l = 0; l < BK; ++l) {
      float a4[R], b4[R];
#pragma unroll
      for (int a = 0; a < R; ++a) { a4[a] = As[l][ti * R + a]; b4[a] = Bs[l][tj * R + a]; }
#pragma unroll
      for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b) acc[a][b] += a4[a] * b4[b];
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < R; ++a) {
    const int gi = i0 + ti * R + a;
    if (gi >= M) continue;
#pragma unroll
    for (int b = 0; b < R; ++b) {
      const int gj = j0 + tj * R + b;
      if (gj >= N) continue;
      float v = acc[a][b] + (bias ? bias[gj] : 0.f);
      float* dst = C + (size_t)gi * ldc + gj;
      *dst = ACC ? *dst + v : v;
    }
  }
}

// C[M, N] (+)= A B over L: 64-edge tiles when they give the chip at least 128 blocks, else 32-edge ones
template <bool ACC>
static void launch_gemm_f32(hipStream_t st, const float* A, long long ai, long long al, const float* Bm, long long bl,
                            long long bj, const float* bias, float* C, int ldc, int M, int N, int L) {
  if ((long long)((M + 63) / 64) * ((N + 63) / 64) >= 128)
    hipLaunchKernelGGL((gemm_f32_kernel<ACC, 64, 32>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, A, ai, al,
                       Bm, bl, bj, bias, C, ldc, M, N, L);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<ACC, 32, 64>), dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, st, A, ai, al,
                       Bm, bl, bj, bias, C, ldc, M, N, L);
}

__global__ __launch_bounds__(256) void colsum_acc_kernel(const float* __restrict__ g, int B, int N,
                                                         float* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += g[(size_t)b * N + n];
  out[n] += s;
}

extern "C" int nbdt_linear_fwd(const float* x, const float* w, const float* b, int32_t B, int32_t K, int32_t N,
                               float* z, void* stream) {
  NBDT_REQUIRE(x && w && z && B > 0 && K > 0 && N > 0, "bad linear arguments");
  if (N >= 64) {
    launch_gemm_f32<false>((hipStream_t)stream, x, (long long)K, 1ll, w, 1ll, (long long)K, b, z, N, B, N, K);
    NBDT_LAUNCH_CHECK();
    return NBDT_OK;
  }
  const long long outs = (long long)B * N;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, B,
                     K, N, z);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_linear_bwd(const float* x, const float* w, const float* gz, int32_t B, int32_t K, int32_t N,
                               float* gx, float* gw, float* gb, void* stream) {
  NBDT_REQUIRE(x && w && gz && B > 0 && K > 0 && N > 0, "bad linear arguments");
  hipStream_t st = (hipStream_t)stream;
  if (N >= 64) {
    if (gx) {   // gx[B,K] = gz[B,N] w[N,K]
      launch_gemm_f32<false>(st, gz, (long long)N, 1ll, w, (long long)K, 1ll, (const float*)nullptr, gx, K, B, K, N);
      NBDT_LAUNCH_CHECK();
    }
    if (gw) {   // gw[N,K] += gz^T[N,B] x[B,K]
      launch_gemm_f32<true>(st, gz, 1ll, (long long)N, x, (long long)K, 1ll, (const float*)nullptr, gw, K, N, K, B);
      NBDT_LAUNCH_CHECK();
      if (gb) {
        hipLaunchKernelGGL(colsum_acc_kernel, dim3((N + 255) / 256), dim3(256), 0, st, gz, B, N, gb);
        NBDT_LAUNCH_CHECK();
      }
    }
    return NBDT_OK;
  }
  if (gx) {
    const long long n = (long long)B * K;
    hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gz, w, B, K, N, gx);
    NBDT_LAUNCH_CHECK();
  }
  if (gw) {
    const long long n = (long long)N * (K + 1);
    const int splits = (B >= 64 && !deterministic()) ? 16 : 1;   // (one batch split: one add per address)
    const int bpb = (B + splits - 1) / splits;
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)((n + 255) / 256), (B + bpb - 1) / bpb), dim3(256), 0, st, gz,
                       x, B, K, N, bpb, gw, gb);
    NBDT_LAUNCH_CHECK();
  }
  return NBDT_OK;
}

// ------------------------------------------------------------------------------------------ sgd
// zero_g: also leave the gradient buffer zeroed for the next step's accumulating weight-gradient kernels (a separate
// 146 MB fill took 108 us per WRN-28-10 step; here it is one more store stream of a pass that is HBM-bound anyway)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g,
                                                  float* __restrict__ buf, long long n, float lr, float momentum,
                                                  float wd, float gscale, bf16_t* __restrict__ pb, int zero_g) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 pv = ((const float4*)p)[i];
    const float4 gv = ((const float4*)g)[i];
    float4 bv = ((const float4*)buf)[i];
    bv.x = momentum * bv.x + (gscale * gv.x + wd * pv.x);
    bv.y = momentum * bv.y + (gscale * gv.y + wd * pv.y);
    bv.z = momentum * bv.z + (gscale * gv.z + wd * pv.z);
    bv.w = momentum * bv.w + (gscale * gv.w + wd * pv.w);
    pv.x -= lr * bv.x; pv.y -= lr * bv.y; pv.z -= lr * bv.z; pv.w -= lr * bv.w;
    ((float4*)buf)[i] = bv;
    ((float4*)p)[i] = pv;
    if (zero_g) ((float4*)g)[i] = float4{0.f, 0.f, 0.f, 0.f};
    if (pb) {
      uint2 o;
      o.x = pack_bf16x2(pv.x, pv.y);
      o.y = pack_bf16x2(pv.z, pv.w);
      ((uint2*)pb)[i] = o;
    }
  }
  // tail (n % 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float b = momentum * buf[i] + (gscale * g[i] + wd * p[i]);
    buf[i] = b;
    p[i] -= lr * b;
    if (zero_g) g[i] = 0.f;
    if (pb) pb[i] = f32_to_bf16(p[i]);
  }
}

extern "C" int nbdt_sgd_step(float* p, float* g, float* buf, int64_t n, float lr, float momentum,
                             float weight_decay, float grad_scale, void* p_bf16, int32_t zero_grad, void* stream) {
  NBDT_REQUIRE(p && g && buf && n > 0, "bad sgd arguments");
  NBDT_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)buf % 16) == 0,
               "flat buffers must be 16-byte aligned");
  const long long n4 = n >> 2;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, (long long)n, lr,
                     momentum, weight_decay, grad_scale, (bf16_t*)p_bf16, zero_grad ? 1 : 0);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}
