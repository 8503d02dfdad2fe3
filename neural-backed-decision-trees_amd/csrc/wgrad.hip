// Convolution weight gradient on MFMA (gfx950), bf16 operands / fp32 accumulate + fp32 atomics.
//
// Replaces cuDNN/MIOpen wgrad behind autograd of nn.Conv2d (reference backbones
// nbdt/models/resnet.py:47-66 and the pytorchcv WRN units of nbdt/models/wideresnet.py:1-5):
//
//     dw[co][w_tap[t]][ci] += sum_m  gy[pix_g(m)][co] * x[pix_x(m) + tap_off[t]][ci]
//
// GEMM view: M = cout, N = cin (per tap), K = pixels (B*gh*gw, up to 2^19 for WRN stage 1).
// Both operands are channel-contiguous in HBM (NHWC) but MFMA wants K(pixel)-contiguous lanes,
// so the TRANSPOSE happens once per staged element, in registers, on the way into LDS:
// each thread loads 4 consecutive pixels x 8 channels (4 x 16 B, full 128-B lines per pixel across
// 8 lanes), transposes the 4x8 bf16 block with v_perm_b32 and writes 8 x ds_write_b64 into a
// [channel][32 pixel] tile (64-B rows, same XOR swizzle and the same ds_read_b128 fragment reads as
// the forward kernel).  Block = NW waves; wave w owns cin tile w and all MT cout tiles.
// The pixel range is split across blocks (grid.x) and partial sums are combined with fp32
// atomics straight into the gradient buffer (which therefore has "+=" semantics like .grad).
//
// Roofline: MFMA-bound, flops = 2*M*cout*cin*ntaps; LDS write port is the secondary limit.
#include "common.h"

using namespace nbdt;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct WgradParams {
  nbdt_wgrad_desc d;
  const bf16_t* x;
  const bf16_t* gy;
  float* dw;
  int M;               // pixels
  int chunks;          // ceil(M / 32)
  int chunks_per_split;
  int n_ci_blocks;     // cin / (32*NW)
  FastDiv div_gw, div_gh;
};

__device__ __forceinline__ int lds_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

template <int MT, int NW>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_kernel(WgradParams p) {
  constexpr int THREADS = 64 * NW;
  constexpr int G_ROWS = 32 * MT, X_ROWS = 32 * NW;
  constexpr int G_BYTES = G_ROWS * 64, X_BYTES = X_ROWS * 64;
  // one task = 4 pixels x 8 channels.  gy tile: 8 pixel groups x 4*MT chunks; x tile: 8 x 4*NW.
  constexpr int G_TASKS = 8 * 4 * MT, X_TASKS = 8 * 4 * NW;
  constexpr int ALL_TASKS = G_TASKS + X_TASKS;
  constexpr int TPT = (ALL_TASKS + THREADS - 1) / THREADS;  // tasks per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [G buf0][X buf0][G buf1][X buf1]

  const nbdt_wgrad_desc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int tap = blockIdx.y;
  const int co_blk = blockIdx.z / p.n_ci_blocks;
  const int ci_blk = blockIdx.z - co_blk * p.n_ci_blocks;
  const int co0 = co_blk * G_ROWS;
  const int ci0 = ci_blk * X_ROWS;
  const int c_begin = blockIdx.x * p.chunks_per_split;
  int c_end = c_begin + p.chunks_per_split;
  c_end = c_end < p.chunks ? c_end : p.chunks;
  if (c_begin >= c_end) return;
  const int x_tap_off = d.tap_off[tap];

  u32x4 rg[TPT][4];

  // task decode (fixed per thread): pixel group fastest so 8 lanes x 16 B cover one pixel's 128 B
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      const int task = tid + q * THREADS;
      if (task >= ALL_TASKS) break;
      const bool is_g = task < G_TASKS;
      const int tt = is_g ? task : task - G_TASKS;
      const int pg = tt & 7;
      const int cc = tt >> 3;
      const int m = chunk * 32 + pg * 4;
      const bool valid = m < p.M;  // M % 4 == 0 (gw % 4 == 0), so a 4-pixel group is all-or-nothing
      const unsigned mm = valid ? (unsigned)m : 0u;
      const unsigned t1 = fdiv(mm, p.div_gw);
      const int j = (int)(mm - t1 * p.div_gw.d);
      const unsigned b = fdiv(t1, p.div_gh);
      const int i = (int)(t1 - b * p.div_gh.d);
      const bf16_t* src;
      int step;
      if (is_g) {
        src = p.gy + ((int)b * d.g_bs + i * d.g_hs + j * d.g_ws + d.g_base + co0 + cc * 8);
        step = d.g_ws;
      } else {
        src = p.x + ((int)b * d.x_bs + i * d.x_hs + j * d.x_ws + d.x_base + x_tap_off + ci0 + cc * 8);
        step = d.x_ws;
      }
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        u32x4 v = *(const u32x4*)(src + px * step);
        if (!valid) v = u32x4{0u, 0u, 0u, 0u};
        rg[q][px] = v;
      }
    }
  };

  auto store_chunk = [&](int buf) {
    unsigned char* base = smem + buf * (G_BYTES + X_BYTES);
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      const int task = tid + q * THREADS;
      if (task >= ALL_TASKS) break;
      const bool is_g = task < G_TASKS;
      const int tt = is_g ? task : task - G_TASKS;
      const int pg = tt & 7;
      const int cc = tt >> 3;
      unsigned char* tile = is_g ? base : base + G_BYTES;
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {  // channel pair (2cp, 2cp+1) of this 8-channel chunk
        const unsigned w0 = rg[q][0][cp], w1 = rg[q][1][cp], w2 = rg[q][2][cp], w3 = rg[q][3][cp];
        u32x2 even, odd;
        even[0] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);  // [p0.lo, p1.lo]
        even[1] = __builtin_amdgcn_perm(w3, w2, 0x05040100u);  // [p2.lo, p3.lo]
        odd[0] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);   // [p0.hi, p1.hi]
        odd[1] = __builtin_amdgcn_perm(w3, w2, 0x07060302u);
        const int row_e = cc * 8 + 2 * cp, row_o = row_e + 1;
        *(u32x2*)(tile + lds_off(row_e, pg >> 1) + (pg & 1) * 8) = even;
        *(u32x2*)(tile + lds_off(row_o, pg >> 1) + (pg & 1) * 8) = odd;
      }
    }
  };

  f32x16 acc[MT];
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;

  const int frag_row = lane & 31, frag_half = lane >> 5;
  auto compute = [&](int buf) {
    const unsigned char* Gs = smem + buf * (G_BYTES + X_BYTES);
    const unsigned char* Xs = Gs + G_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 2 * ks + frag_half;
      const bf16x8 xf = *(const bf16x8*)(Xs + lds_off(wave * 32 + frag_row, c));
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) {
        const bf16x8 gf = *(const bf16x8*)(Gs + lds_off(tm * 32 + frag_row, c));
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, xf, acc[tm], 0, 0, 0);
      }
    }
  };

  load_chunk(c_begin);
  store_chunk(0);
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int cur = (c - c_begin) & 1;
    const bool more = c + 1 < c_end;
    if (more) load_chunk(c + 1);
    compute(cur);
    if (more) store_chunk(cur ^ 1);
    __syncthreads();
  }

  // epilogue: lane owns ci = ci0 + 32*wave + (lane&31); regs run over co
  const int ci = ci0 + wave * 32 + frag_row;
  const int w_tap = d.w_tap[tap];
#pragma unroll
  for (int tm = 0; tm < MT; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * frag_half;
      atomicAdd(p.dw + ((int64_t)co * d.w_ntaps + w_tap) * d.cin + ci, acc[tm][r]);
    }
}

template <int MT, int NW>
static int launch(WgradParams& p, hipStream_t st) {
  const nbdt_wgrad_desc& d = p.d;
  p.n_ci_blocks = d.cin / (32 * NW);
  const int tiles = d.ntaps * (d.cout / (32 * MT)) * p.n_ci_blocks;
  // split the pixel range so that ~3 blocks per CU are in flight, but keep >= 8 chunks per block
  int splits = (768 + tiles - 1) / tiles;
  const int max_splits = p.chunks / 8 > 0 ? p.chunks / 8 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.chunks_per_split = (p.chunks + splits - 1) / splits;
  splits = (p.chunks + p.chunks_per_split - 1) / p.chunks_per_split;
  const size_t shmem = 2 * (size_t)(32 * MT + 32 * NW) * 64;
  hipLaunchKernelGGL((conv_wgrad_kernel<MT, NW>), dim3(splits, d.ntaps, (d.cout / (32 * MT)) * p.n_ci_blocks),
                     dim3(64 * NW), shmem, st, p);
  NBDT_LAUNCH_CHECK();
  return NBDT_OK;
}

extern "C" int nbdt_conv_wgrad(const nbdt_wgrad_desc* d, const void* x, const void* gy, float* dw,
                               void* stream) {
  NBDT_REQUIRE(d && x && gy && dw, "null argument");
  NBDT_REQUIRE(d->cin > 0 && d->cin % 32 == 0 && d->cout > 0 && d->cout % 32 == 0, "channels must be multiples of 32");
  NBDT_REQUIRE(d->ntaps >= 1 && d->ntaps <= 9 && d->w_ntaps >= 1, "bad tap table");
  NBDT_REQUIRE(d->B > 0 && d->gh > 0 && d->gw > 0 && d->gw % 4 == 0, "pixel grid width must be a multiple of 4");
  for (int t = 0; t < d->ntaps; ++t) {
    NBDT_REQUIRE(d->w_tap[t] >= 0 && d->w_tap[t] < d->w_ntaps, "bad w_tap");
    NBDT_REQUIRE(d->tap_off[t] % 8 == 0, "tap offsets must be 16-byte aligned");
  }
  NBDT_REQUIRE(d->x_bs % 8 == 0 && d->x_hs % 8 == 0 && d->x_ws % 8 == 0 && d->x_base % 8 == 0 &&
               d->g_bs % 8 == 0 && d->g_hs % 8 == 0 && d->g_ws % 8 == 0 && d->g_base % 8 == 0,
               "pixel offsets must be 16-byte aligned");
  WgradParams p;
  p.d = *d;
  p.x = (const bf16_t*)x;
  p.gy = (const bf16_t*)gy;
  p.dw = dw;
  const int64_t M64 = (int64_t)d->B * d->gh * d->gw;
  NBDT_REQUIRE(M64 < (1ll << 31), "pixel grid too large");
  p.M = (int)M64;
  p.chunks = (p.M + 31) / 32;
  p.div_gw = make_fastdiv((unsigned)d->gw);
  p.div_gh = make_fastdiv((unsigned)d->gh);
  hipStream_t st = (hipStream_t)stream;
  const int mt32 = d->cout / 32, nt32 = d->cin / 32;
  // cout tile: 160 (WRN widths) or 128/64/32; cin tile (= waves): 5, 4, 2 or 1
#define NBDT_WG(MT_, NW_) return launch<MT_, NW_>(p, st)
  if (mt32 % 5 == 0) {
    if (nt32 % 5 == 0) NBDT_WG(5, 5);
    if (nt32 % 4 == 0) NBDT_WG(5, 4);
    if (nt32 % 2 == 0) NBDT_WG(5, 2);
    NBDT_WG(5, 1);
  }
  if (mt32 % 4 == 0) {
    if (nt32 % 4 == 0) NBDT_WG(4, 4);
    if (nt32 % 2 == 0) NBDT_WG(4, 2);
    NBDT_WG(4, 1);
  }
  if (mt32 % 2 == 0) {
    if (nt32 % 4 == 0) NBDT_WG(2, 4);
    if (nt32 % 2 == 0) NBDT_WG(2, 2);
    NBDT_WG(2, 1);
  }
  if (nt32 % 4 == 0) NBDT_WG(1, 4);
  if (nt32 % 2 == 0) NBDT_WG(1, 2);
  NBDT_WG(1, 1);
#undef NBDT_WG
}
