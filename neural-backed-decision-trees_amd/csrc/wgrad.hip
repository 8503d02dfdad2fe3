// Convolution weight gradient on MFMA (gfx950), bf16 operands / fp32 accumulate + fp32 atomics.
//
// Replaces cuDNN/MIOpen wgrad behind autograd of nn.Conv2d (reference backbones
// nbdt/models/resnet.py:47-66 and the pytorchcv WRN units of nbdt/models/wideresnet.py:1-5):
//
//     dw[co][w_tap[t]][ci] += sum_m  gy[pix_g(m)][co] * x[pix_x(m) + tap_off[t]][ci]
//
// GEMM view: M = cout, N = cin (per tap), K = pixels (B*gh*gw, up to 2^19 for WRN stage 1).
// Both operands are channel-contiguous in HBM (NHWC) but MFMA wants K(pixel)-contiguous lanes, so
// the TRANSPOSE happens once per staged element, in registers, on the way into LDS: a thread loads
// 4 consecutive pixels x 8 channels (4 x 16 B; consecutive lanes take consecutive channel chunks
// of the same pixels, i.e. whole 128-B lines), transposes the 4x8 bf16 block with v_perm_b32 and
// writes 8 x ds_write_b64 into a [channel][64 pixel] tile (128-B rows).
//
// v2 structure (v1 was latency-bound at 8% of MFMA peak: 10 MFMAs per barrier, 1-deep prefetch):
//   * block = 4 waves as 2(co) x 2(ci); wave tile = (16*WM) x (16*WN) out of v_mfma_f32_16x16x32
//     (WM = WN = 5 -> 160 x 160 block for the WideResNet widths; 4/2/1 for power-of-two widths);
//   * K chunk = 64 pixels per barrier: 2*WM*WN MFMAs per wave between barriers (50 for 5x5);
//   * register prefetch: the loads of chunk t+1 are issued before the MFMAs of chunk t and
//     transposed into the other LDS buffer after them (2 blocks/CU interleave the rest);
//   * LDS rows are 128 B with chunk' = chunk ^ (((row>>1) ^ (row>>3)) & 7): conflict-free for
//     the 16-row ds_read_b128 fragment reads and 2-way at worst for the transposing writes.
// The pixel range is split across blocks (grid.x) and partial sums are combined with fp32
// atomics straight into the gradient buffer (which therefore has "+=" semantics like .grad).
//
// Roofline: MFMA-bound, flops = 2*M*cout*cin*ntaps.
#include "common.h"
#include <stdlib.h>

using namespace nbdt;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct WgradParams {
  nbdt_wgrad_desc d;
  const bf16_t* x;
  const bf16_t* gy;
  float* dw;
  int M;               // pixels
  int chunks;          // ceil(M / 64)
  int chunks_per_split;
  int n_ci_blocks;     // cin / (32*WN)
  int splits, items, per_xcd;
  FastDiv div_gw, div_gh;
};

constexpr int KC = 64;  // pixels per K chunk

// byte offset of (row, 16-byte chunk c in 0..7) inside a [rows][64 px] bf16 tile (128-B rows)
__device__ __forceinline__ int d_ntaps(const WgradParams& p) { return p.d.ntaps; }

__device__ __forceinline__ int wg_off(int row, int c) {
  return row * 128 + ((c ^ (((row >> 1) ^ (row >> 3)) & 7)) << 4);
}

template <int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
  constexpr int G_ROWS = 32 * WM, X_ROWS = 32 * WN;           // block tile: couts x cins
  constexpr int G_BYTES = G_ROWS * 128, X_BYTES = X_ROWS * 128;
  constexpr int STAGE = G_BYTES + X_BYTES;
  // one task = 4 pixels x 8 channels; channel chunk fastest across lanes (coalesced lines)
  constexpr int G_CH = G_ROWS / 8, X_CH = X_ROWS / 8;
  constexpr int G_TASKS = 16 * G_CH, X_TASKS = 16 * X_CH;
  constexpr int ALL_TASKS = G_TASKS + X_TASKS;
  constexpr int TPT = (ALL_TASKS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [stage0][stage1]

  const nbdt_wgrad_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware work order: block b runs on XCD b%8; each XCD walks a contiguous range of items and
  // the tap index is the FASTEST item coordinate, so the 9 tap-blocks that re-read the same
  // gy / x pixel range run back-to-back on ONE XCD and share its private L2 (without this every tap
  // streams both tensors from HBM/MALL again: 9x the traffic, measured 8% of MFMA peak).
  const int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if (item >= p.items) return;
  const int tap = item % d_ntaps(p);
  const int rest = item / d_ntaps(p);
  const int split = rest % p.splits;
  const int tile = rest / p.splits;
  const int co_blk = tile / p.n_ci_blocks;
  const int ci_blk = tile - co_blk * p.n_ci_blocks;
  const int co0 = co_blk * G_ROWS;
  const int ci0 = ci_blk * X_ROWS;
  const int c_begin = split * p.chunks_per_split;
  int c_end = c_begin + p.chunks_per_split;
  c_end = c_end < p.chunks ? c_end : p.chunks;
  if (c_begin >= c_end) return;
  const int x_tap_off = d.tap_off[tap];

  // ---- per-thread task table (fixed for the whole kernel)
  int t_pg[TPT], t_row0[TPT], t_choff[TPT], t_step[TPT], t_zero[TPT];
  bool t_isg[TPT], t_on[TPT];
#pragma unroll
  for (int q = 0; q < TPT; ++q) {
    const int task = tid + q * 256;
    t_on[q] = task < ALL_TASKS;
    const bool is_g = task < G_TASKS;
    const int tt = is_g ? task : task - G_TASKS;
    const int nch = is_g ? G_CH : X_CH;
    const int cc = tt % nch;
    t_pg[q] = tt / nch;
    t_isg[q] = is_g;
    t_row0[q] = cc * 8;
    t_choff[q] = is_g ? (d.g_base + co0 + cc * 8) : (d.x_base + x_tap_off + ci0 + cc * 8);
    t_step[q] = is_g ? d.g_ws : d.x_ws;
    // 4 consecutive zero pixels: the top border row of image 0 (row length (W+2)*C >= 4*step)
    t_zero[q] = t_on[q] ? (is_g ? co0 + cc * 8 : ci0 + cc * 8) : 0;
    if (!t_on[q]) { t_pg[q] = 0; t_choff[q] = 0; }
  }

  // Loads are UNCONDITIONAL and their results are not touched until store_chunk, so the compiler
  // keeps all 4*TPT loads in flight across the MFMA block (a select or a branch on the loaded
  // value makes hipcc wait vmcnt(0) right behind each load: measured 8% of MFMA peak).
  // Out-of-range pixel groups (K tail) and idle task slots read the tensor's top-left BORDER
  // pixels instead, which are zero by construction of the padded layout.
  auto load_chunk = [&](u32x4 (&rg)[TPT][4], int chunk) {
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      const int m = chunk * KC + t_pg[q] * 4;
      const bool valid = t_on[q] && m < p.M;  // M % 4 == 0: a 4-pixel group is all-or-nothing
      const unsigned mm = valid ? (unsigned)m : 0u;
      const unsigned t1 = fdiv(mm, p.div_gw);
      const int j = (int)(mm - t1 * p.div_gw.d);
      const unsigned b = fdiv(t1, p.div_gh);
      const int i = (int)(t1 - b * p.div_gh.d);
      int off = t_isg[q] ? ((int)b * d.g_bs + i * d.g_hs + j * d.g_ws + t_choff[q])
                         : ((int)b * d.x_bs + i * d.x_hs + j * d.x_ws + t_choff[q]);
      off = valid ? off : t_zero[q];
      const bf16_t* src = (t_isg[q] ? p.gy : p.x) + off;
#pragma unroll
      for (int px = 0; px < 4; ++px) rg[q][px] = *(const u32x4*)(src + px * t_step[q]);
    }
  };

  auto store_chunk = [&](const u32x4 (&rg)[TPT][4], int buf) {
    unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      if (!t_on[q]) continue;
      unsigned char* tile = t_isg[q] ? base : base + G_BYTES;
      const int pg = t_pg[q];
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {  // channel pair (2cp, 2cp+1) of this 8-channel chunk
        const unsigned w0 = rg[q][0][cp], w1 = rg[q][1][cp], w2 = rg[q][2][cp], w3 = rg[q][3][cp];
        u32x2 even, odd;
        even[0] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);  // [p0.lo, p1.lo]
        even[1] = __builtin_amdgcn_perm(w3, w2, 0x05040100u);  // [p2.lo, p3.lo]
        odd[0] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);   // [p0.hi, p1.hi]
        odd[1] = __builtin_amdgcn_perm(w3, w2, 0x07060302u);
        const int row_e = t_row0[q] + 2 * cp, row_o = row_e + 1;
        *(u32x2*)(tile + wg_off(row_e, pg >> 1) + (pg & 1) * 8) = even;
        *(u32x2*)(tile + wg_off(row_o, pg >> 1) + (pg & 1) * 8) = odd;
      }
    }
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15, frag_k = lane >> 4;
  auto compute = [&](int buf) {
    const unsigned char* Gs = smem + buf * STAGE;
    const unsigned char* Xs = Gs + G_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 4 * ks + frag_k;
      bf16x8 gf[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a) gf[a] = *(const bf16x8*)(Gs + wg_off((wm * WM + a) * 16 + frag_row, c));
#pragma unroll
      for (int b = 0; b < WN; ++b) {
        const bf16x8 xf = *(const bf16x8*)(Xs + wg_off((wn * WN + b) * 16 + frag_row, c));
#pragma unroll
        for (int a = 0; a < WM; ++a)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[a], xf, acc[a][b], 0, 0, 0);
      }
    }
  };

  // ---- main loop: loads of chunk t+1 are issued before the MFMAs of chunk t and transposed into
  // the other LDS buffer after them; one barrier per chunk (a 2-deep register ring spills at
  // 5x5 tiles: 100 accumulators + 2 x 48 staging registers)
  u32x4 rg[TPT][4];
  load_chunk(rg, c_begin);
  store_chunk(rg, 0);
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int cur = (c - c_begin) & 1;
    const bool more = c + 1 < c_end;
    if (more) load_chunk(rg, c + 1);
    compute(cur);
    if (more) store_chunk(rg, cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: 16x16 tile, lane: ci = col (lane&15), regs r -> co = 4*(lane>>4) + r
  const int w_tap = d.w_tap[tap];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const int ci = ci0 + (wn * WN + b) * 16 + frag_row;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wm * WM + a) * 16 + 4 * frag_k + r;
        atomicAdd(p.dw + ((int64_t)co * d.w_ntaps + w_tap) * d.cin + ci, acc[a][b][r]);
      }
    }
}

template <int WM, int WN>
static int launch(WgradParams& p, hipStream_t st) {
  const nbdt_wgrad_desc& d = p.d;
  p.n_ci_blocks = d.cin / (32 * WN);
  const int tiles = d.ntaps * (d.cout / (32 * WM)) * p.n_ci_blocks;
  // split the pixel range so that ~2 blocks per CU are in flight, but keep >= 8 chunks per block
  int splits = (512 + tiles - 1) / tiles;
  const int max_splits = p.chunks / 8 > 0 ? p.chunks / 8 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.chunks_per_split = (p.chunks + splits - 1) / splits;
  splits = (p.chunks + p.chunks_per_split - 1) / p.chunks_per_split;
  const size_t shmem = 2 * (size_t)(32 * WM + 32 * WN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    NBDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<WM, WN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_set = true;
  }
  p.splits = splits;
  p.items = tiles * splits;
  p.per_xcd = (p.items + 7) / 8;
  hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN>), dim3(p.per_xcd * 8), dim3(256), shmem, st, p);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_conv_wgrad(const nbdt_wgrad_desc* d, const void* x, const void* gy, float* dw,
                               void* stream) {
  NBDT_REQUIRE(d && x && gy && dw, "null argument");
  NBDT_REQUIRE(d->cin > 0 && d->cin % 32 == 0 && d->cout > 0 && d->cout % 32 == 0, "channels must be multiples of 32");
  NBDT_REQUIRE(d->ntaps >= 1 && d->ntaps <= 9 && d->w_ntaps >= 1, "bad tap table");
  NBDT_REQUIRE(d->B > 0 && d->gh > 0 && d->gw > 0, "empty pixel grid");
  for (int t = 0; t < d->ntaps; ++t) {
    NBDT_REQUIRE(d->w_tap[t] >= 0 && d->w_tap[t] < d->w_ntaps, "bad w_tap");
    NBDT_REQUIRE(d->tap_off[t] % 8 == 0, "tap offsets must be 16-byte aligned");
  }
  NBDT_REQUIRE(d->x_bs % 8 == 0 && d->x_hs % 8 == 0 && d->x_ws % 8 == 0 && d->x_base % 8 == 0 &&
               d->g_bs % 8 == 0 && d->g_hs % 8 == 0 && d->g_ws % 8 == 0 && d->g_base % 8 == 0,
               "pixel offsets must be 16-byte aligned");
  const int64_t M_all = (int64_t)d->B * d->gh * d->gw;
  NBDT_REQUIRE(M_all < (1ll << 31), "pixel grid too large");
  // default: v3 (LDS-DMA + transpose reads, wgrad_dma.hip).  NBDT_WGRAD_V2=1 selects the
  // register-staged v2 kernel below (kept for A/B measurements; needs gw % 4 == 0).
  static const bool use_v2 = getenv("NBDT_WGRAD_V2") != nullptr;
  if (!use_v2) {
    static const bool no_taps = getenv("NBDT_NO_TAPS") != nullptr;
    if (!no_taps && nbdt::wgrad_taps_applicable(d)) return nbdt::wgrad_taps(d, x, gy, dw, (hipStream_t)stream);
    return nbdt::wgrad_dma(d, x, gy, dw, (hipStream_t)stream);
  }
  NBDT_REQUIRE(d->gw % 4 == 0, "v2 wgrad: pixel grid width must be a multiple of 4");
  WgradParams p;
  p.d = *d;
  p.x = (const bf16_t*)x;
  p.gy = (const bf16_t*)gy;
  p.dw = dw;
  const int64_t M64 = (int64_t)d->B * d->gh * d->gw;
  NBDT_REQUIRE(M64 < (1ll << 31), "pixel grid too large");
  p.M = (int)M64;
  p.chunks = (p.M + KC - 1) / KC;
  p.div_gw = make_fastdiv((unsigned)d->gw);
  p.div_gh = make_fastdiv((unsigned)d->gh);
  hipStream_t st = (hipStream_t)stream;
  const int mt = d->cout / 32, nt = d->cin / 32;
  // block tile (32*WM couts) x (32*WN cins): 160 for the WRN widths, else 128 / 64 / 32
#define NBDT_WG(WM_, WN_) return launch<WM_, WN_>(p, st)
#define NBDT_WG_ROW(WM_)                 \
  do {                                   \
    if (nt % 5 == 0) NBDT_WG(WM_, 5);    \
    if (nt % 4 == 0) NBDT_WG(WM_, 4);    \
    if (nt % 2 == 0) NBDT_WG(WM_, 2);    \
    NBDT_WG(WM_, 1);                     \
  } while (0)
  if (mt % 5 == 0) NBDT_WG_ROW(5);
  if (mt % 4 == 0) NBDT_WG_ROW(4);
  if (mt % 2 == 0) NBDT_WG_ROW(2);
  NBDT_WG_ROW(1);
#undef NBDT_WG_ROW
#undef NBDT_WG
}
