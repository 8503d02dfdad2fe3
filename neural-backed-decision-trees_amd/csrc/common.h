// Shared helpers for libnbdt_hip.so (gfx950 only -- no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "../../include/nbdt_hip.h"

// ---- timing-experiment switches.  A handful of -D switches in the kernel sources skip, fake or re-time part of a kernel
// (ablations, "what would X buy" experiments, s_memtime stamps): several of them produce WRONG results by design.  They
// compile only together with -DNBDT_TIMING_BUILD (scratch/build_variants.sh passes it), and such an object exports
// nbdt_timing_build, which makes nbdt/_C.py refuse the library unless NBDT_ALLOW_TIMING_BUILD=1 -- a stray -D in a
// product build is a compile error, not a silently wrong gradient.
#if !defined(NBDT_TIMING_BUILD) &&                                                                                    \
    (defined(NBDT_WPP_NO_EPI) || defined(NBDT_WKS_NO_EXCHANGE) || defined(NBDT_WKS_DMA_IN_M) || defined(NBDT_WPP_MIN_STAGES) || defined(NBDT_WPP_FRAC8) || defined(NBDT_PP_KFRAC5) ||    \
     defined(NBDT_DMA_WTILED_FAKE) || defined(NBDT_PP_DUMMY_VALU) || defined(NBDT_PP_NO_PERSIST) || defined(NBDT_PP_NO_PAD) ||                    \
     defined(NBDT_HALO_NO_ACCUMULATE) || defined(NBDT_DW_TARGET) || defined(NBDT_DW_U) || defined(NBDT_HEAD_SPB) ||    \
     defined(NBDT_EPI_TIMING) || defined(NBDT_EPI_STATS_ATOMICS) || defined(NBDT_WGT_NSTAGE) || defined(NBDT_NO_XCD_CONTIGUOUS) || defined(NBDT_CUS_IN_FLIGHT) || defined(NBDT_NT_MIN_MB) || defined(NBDT_PLAIN_STORES) ||                      \
     (defined(NBDT_HEAD_SKIP) && (NBDT_HEAD_SKIP + 0) != 0) || (defined(NBDT_PP_ABLATE) && (NBDT_PP_ABLATE + 0) != 0) || \
     (defined(NBDT_PP_SCHED) && (NBDT_PP_SCHED + 0) != 0) || (defined(NBDT_PP_TIMING) && (NBDT_PP_TIMING + 0) != 0) ||  \
     (defined(NBDT_WPP_TIMING) && (NBDT_WPP_TIMING + 0) != 0) || (defined(NBDT_SEG_TIMING) && (NBDT_SEG_TIMING + 0) != 0) || (defined(NBDT_RULES_TIMING) && (NBDT_RULES_TIMING + 0) != 0))
#error "timing-experiment switch without -DNBDT_TIMING_BUILD: these switches change what the kernels compute (csrc/common.h)"
#endif
#ifdef NBDT_TIMING_BUILD
extern "C" __attribute__((weak, used, visibility("default"))) int nbdt_timing_build() { return 1; }
#endif

namespace nbdt {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return code;
}

#define NBDT_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) return nbdt::fail(NBDT_EHIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define NBDT_REQUIRE(cond, msg)                                       \
  do {                                                                \
    if (!(cond)) return nbdt::fail(NBDT_EINVAL, "%s (%s)", msg, #cond); \
  } while (0)

#define NBDT_LAUNCH_CHECK() NBDT_HIP_CHECK(hipGetLastError())

// hipFuncSetAttribute acts on the CURRENT device's copy of a kernel, so "the attribute is set" is a per-device
// fact.  One DeviceAttr per launch site: need(bytes) returns true (holding the site's mutex until done()) when the
// current device has not been configured for at least `bytes` of dynamic LDS yet.  Thread-safe; ~20 ns per launch
// once configured.
struct DeviceAttr {
  static constexpr int kMaxDevices = 64;
  std::mutex m;
  size_t bytes[kMaxDevices] = {};
  int dev = 0;
  bool need(size_t want) {
    int d = 0;
    (void)hipGetDevice(&d);
    if (d < 0 || d >= kMaxDevices) d = kMaxDevices - 1;
    m.lock();
    dev = d;                     // (owned by whoever holds the mutex)
    if (bytes[d] >= want && want > 0) { m.unlock(); return false; }
    return true;                 // caller sets the attributes, then calls done(want)
  }
  void done(size_t want) { bytes[dev] = want ? want : 1; m.unlock(); }
  void abort() { m.unlock(); }
};
// NBDT_HIP_CHECK for use between need() and done(): releases the site's mutex on failure
#define NBDT_ATTR_CHECK(site, expr)                                                       \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) { (site).abort(); return nbdt::fail(NBDT_EHIP, "%s: %s", #expr, hipGetErrorString(_e)); } \
  } while (0)

// ---- deterministic mode (nbdt_set_deterministic; misc.hip).  Every cross-block fp32 reduction of the backbone
// path normally goes through atomics (32 replicated BatchNorm slots, split-K weight gradients, LDS atomics in the conv
// epilogue), whose order changes from run to run.  With the switch on, each of them adds into a zeroed library-owned
// row per block / per pixel split instead -- one add per address -- and det_fold() sums the rows in index order.
bool deterministic();
// K-split weight gradient: plain stores into per-split copies + det_fold instead of fp32 atomics into dw (default on; deterministic mode: always)
bool wgrad_store_epilogue();
// CUs the one-block-per-CU MFMA kernels leave free (nbdt_set_reserved_cus): a collective's kernels (RCCL, one block per
// channel) running beside the backward pass get them, instead of making persistent blocks wait for a CU they hold
int reserved_cus();
// stream-ordered workspace of at least `floats` floats, one per (device, stream); nullptr if it cannot be allocated
float* det_rows(hipStream_t st, size_t floats);
// why the calling thread's last det_rows() returned nullptr (for the caller's error message)
const char* det_rows_why();
// dst[i] += rows[0][i] + rows[1][i] + ... (ascending r, one thread per i), i < n
int det_fold(hipStream_t st, const float* rows, int nrows, size_t n, float* dst);

typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
// round-to-nearest-even, NaN preserved
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// two fp32 -> packed bf16x2 in ONE instruction (gfx950 v_cvt_pk_bf16_f32: round-to-nearest-even, NaN
// preserved; bit-exactness vs torch's .to(bfloat16) is asserted by tests/test_backbone_gpu.py::test_sgd_*).
// The integer formulation above costs ~10 VALU + a divergent NaN branch per element.
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// exact n / d for 0 <= n < 2^31 with one mul-hi and one shift (host builds, device divides)
struct FastDiv {
  unsigned mul, sh, d;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  if (d == 1) { f.mul = 0; f.sh = 0; return f; }
  unsigned s = 0;
  while ((1ull << s) < d) ++s;  // s = ceil(log2 d)
  const unsigned long long num = 1ull << (31 + s);
  f.mul = (unsigned)((num + d - 1) / d);
  f.sh = s - 1;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.sh);
}

// geometry of a padded NHWC activation tensor [B][H+2][W+2][C] (bf16), interior pixels only
struct PadGeom {
  int B, H, W, C;
  int npix;          // B*H*W
  int row;           // (W+2)*C elements
  int img;           // (H+2)*(W+2)*C elements
  FastDiv div_w, div_h;
};
inline PadGeom make_geom(int B, int H, int W, int C) {
  PadGeom g;
  g.B = B; g.H = H; g.W = W; g.C = C;
  g.npix = B * H * W;
  g.row = (W + 2) * C;
  g.img = (H + 2) * g.row;
  g.div_w = make_fastdiv((unsigned)W);
  g.div_h = make_fastdiv((unsigned)H);
  return g;
}
// element offset of channel 0 of interior pixel p (p enumerates (b, h, w) row-major)
__device__ __forceinline__ int pad_offset(const PadGeom& g, int p) {
  const unsigned t = fdiv((unsigned)p, g.div_w);
  const int w = p - (int)t * g.W;
  const unsigned b = fdiv(t, g.div_h);
  const int h = (int)t - (int)b * g.H;
  return (int)b * g.img + (h + 1) * g.row + (w + 1) * g.C;
}

// wgrad_dma.hip: LDS-DMA + transpose-read weight-gradient kernel (default path of nbdt_conv_wgrad)
int wgrad_dma(const nbdt_wgrad_desc* d, const void* x, const void* gy, float* dw, hipStream_t st);

// wgrad_taps.hip: all-nine-taps weight-gradient kernel for dense 3x3 stride-1 convs (preferred)
bool wgrad_taps_applicable(const nbdt_wgrad_desc* d);
int wgrad_taps(const nbdt_wgrad_desc* d, const void* x, const void* gy, float* dw, hipStream_t st);
int wgrad_taps_blocks(const nbdt_wgrad_desc* d);   // blocks of the 8-wave launch (cu_budget applied), else 0

// wgrad_s2d.hip: 3x3 / stride-2 weight gradient over the space-to-depth copy of the input (ops.conv_wgrad_desc_s2d)
bool wgrad_s2d_applicable(const nbdt_wgrad_desc* d);
int wgrad_s2d(const nbdt_wgrad_desc* d, const void* x, const void* gy, float* dw, hipStream_t st);
int wgrad_s2d_blocks(const nbdt_wgrad_desc* d);     // blocks (= CUs) of that launch, cu_budget applied

// conv_dma.hip: LDS-DMA pipelined implicit GEMM (default path of nbdt_conv_igemm)
struct BnBwdArgs {   // epilogue extras: the BatchNorm whose input gradient a dgrad launch produces (STATS mode 2)
  const void* x;
  const float *mean, *rstd, *gamma, *beta;
  // or (mode 3, inference) a folded eval-mode BatchNorm + activation applied to the conv output
  const float *aff_scale = nullptr, *aff_shift = nullptr;
  int aff_act = 0;
};
int conv_igemm_dma(const nbdt_conv_desc* d, const void* in, const void* w, void* out, const void* res,
                   float* stats, const BnBwdArgs* bn, int M, hipStream_t st);
int conv_igemm_dma_multi(const nbdt_conv_desc* descs, int n, const void* in, const void* w, void* out, hipStream_t st);

// conv_halo.hip: 3x3 stride-1 kernel with an LDS-resident halo tile (preferred when applicable)
struct HaloGeom {
  int ib, rb;          // images per block, rows per image per block  (ib*rb*gw == 256)
  int hw2;             // gw + 2
  int himg;            // (rb+2) * (gw+2): halo pixels per image
  int hp;              // ib * himg
  int a_instr;         // ceil(hp*4 / 64) wave-instructions per halo tile (wave w issues ids w, w+4, ..)
  int a_bytes;         // a_instr * 1024
  int blocks_per_img;  // gh / rb when ib == 1
  int nwv;             // waves per block: 4 (256-pixel tile) or 8 (512-pixel tile, or 256-pixel half tile: mw == 1)
  int mw;              // 32-pixel fragments per wave: 2, or 1 for the 8-wave kernel's half tile
  // LDS image of the halo (conv3x3_pp_kernel): rows of `lpitch` pixel slots.  pad = lpitch - hw2 is 0 (the halo is copied
  // as the one contiguous run it is in memory; a_instr covers hp pixels) or 2 for images narrower than 32 pixels: rows of
  // gw + 4 slots, the last two of each row unused, so that a 32-pixel fragment -- 2 or 4 image rows -- meets every bank
  // once (conv_halo.hip, "LDS pitch").  dv = lpitch - (row multiplier of the swizzle coordinate): 0 or 4.
  int lpitch, limg;    // slots per halo row / per image ((rb+2) * lpitch)
  int pad, dv;
  int row_magic;       // ceil(65536 / lpitch): slot / lpitch == (slot * row_magic) >> 16 for slot < 2048
};
bool conv_halo_applicable(const nbdt_conv_desc* d, int M, HaloGeom* hg);
int conv_ksplit_rule(const nbdt_conv_desc& d, int nt, int items);      // blocks per half tile the launch rule asks for
int conv_halo_items(const nbdt_conv_desc& d, const HaloGeom& hg, int M, int* nt_out);   // (cout tile, items) of a launch
extern thread_local const char* g_last_wgrad;   // ... and the last nbdt_conv_wgrad call
extern thread_local const char* g_last_igemm;   // name of the kernel the last nbdt_conv_igemm* call launched (tests)
extern thread_local char g_last_igemm_full[128];  // ... with its template arguments, as a profiler prints it (bench.py's traffic guard)
struct BnBwdArgs;
int conv3x3_halo(const nbdt_conv_desc* d, const HaloGeom& hg, const void* in, const void* w, void* out,
                 const void* res, float* stats, const BnBwdArgs* bn, int M, hipStream_t st);

}  // namespace nbdt

// LDS-DMA of 16 B per lane (1 KiB per wave) with the address split as <SGPR base> + <32-bit per-lane byte offset>:
// the wave-uniform part of the address stays scalar arithmetic and the lane part is one VGPR.  The pad before the
// load covers "VALU wrote the SGPR (readfirstlane / readlane) -> VMEM reads it as base" (5 wait states; hipcc does
// not pad inside an asm string).  M0 = LDS destination of lane 0 (the image is lane-linear: lane L lands at +16 L).
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 2\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// The same with every wave-uniform operand passed through v_readfirstlane first.  hipcc's instruction selection hands an
// "s"-constrained asm operand a VGPR when its divergence analysis is not sure the value is uniform -- which happens as
// soon as a vector expression of the same function shares a subterm with the scalar address arithmetic (round 4's
// swizzle draft stopped there: "invalid operand for instruction: s_mov_b32 m0, v6").  A readfirstlane of a value that IS
// in an SGPR folds away; one of a value the compiler moved to a VGPR is one VALU instruction and the 5 wait states the pad covers.
__device__ __forceinline__ void glds16_sf(const void* sbase, unsigned voff, unsigned lds_dst) {
  const unsigned long long b = (unsigned long long)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  glds16_s((const void*)(((unsigned long long)hi << 32) | lo), voff, __builtin_amdgcn_readfirstlane(lds_dst));
}

namespace nbdt {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// 16-byte load of an operand that is STREAMED -- read once by this launch, not again soon: the elementwise / reduction
// passes over activation and gradient tensors.  NT: nontemporal, so the stream does not push what the MFMA kernels re-read
// (weight tiles, halo rows) out of the L2s and the Infinity Cache.  A compile-time choice (a run-time select between the two
// load forms measured like plain loads), made per launch by stream_nt(): only tensors too large to stay in the 256 MB cache
// anyway.  WRN-28-10 at 512 images (189 / 106 MB tensors): 16.37 -> 16.23 ms per step; ResNet18 at 128 x 64 x 64 (<= 71 MB)
// and EfficientNet-B0 LOSE 0.4 / 2 % with nontemporal loads -- their consumers find the producers' tensors in the cache
// (profiles/r06_nt_loads_ab.txt).  Nontemporal STORES measured slower and are not used.
template <bool NT>
__device__ __forceinline__ u32x4_t ld16_stream(const void* p) {
  if (NT) return __builtin_nontemporal_load((const u32x4_t*)p);
  return *(const u32x4_t*)p;
}
#ifndef NBDT_NT_MIN_MB
#define NBDT_NT_MIN_MB 96
#endif
inline bool stream_nt(long long tensor_bytes) { return tensor_bytes >= ((long long)NBDT_NT_MIN_MB << 20); }

// 16-byte WRITE-THROUGH store (sc1: the line leaves the XCD's L2 as it is written).  Plain stores leave up to 32 MB dirty in
// the L2s, and the kernel boundary that follows writes them back before the next kernel starts (MI355X_MICROARCH.md,
// "boundary": + B / 6 TB/s) -- 3-5 us per launch that a short consumer kernel pays in full.  Nothing the next kernel reads
// would have hit these lines anyway (another XCD's L2 is not coherent with it).  st16<false> is the plain store (timing A/B).
template <bool WT = true>
__device__ __forceinline__ void st16(void* p, const u32x4_t v) {
#ifdef NBDT_PLAIN_STORES          // timing-only builds: rounds 1-5
  *(u32x4_t*)p = v;
#else
  // (s_nop: a VALU write of the data registers within two wait states of a > 8-byte store corrupts it, and hipcc's hazard
  //  recogniser does not look inside an asm string -- without it the BatchNorm passes lost 2 % of their stores)
  if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
  else *(u32x4_t*)p = v;
#endif
}

__device__ __forceinline__ void unpack8(const u32x4_t v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
  u32x4_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
  return v;
}

}  // namespace nbdt
