// Convolution weight gradient on MFMA (gfx950): C-ABI entry point and dispatch.
//
//     dw[n][w_tap[t]][c] += sum_m  gy[pix_out(m)][n] * x[pix_in(m) + tap_off[t]][c]
//
// Replaces the weight-gradient half of cuDNN/MIOpen's conv backward behind nn.Conv2d (reference
// nbdt/models/resnet.py:47-66, pytorchcv WRN PreResUnit convs via nbdt/models/wideresnet.py:1-5).  K = pixels (up to
// 2^19) and both operands are channel-contiguous, so one of them must be transposed on the way to the MFMA: the
// kernels stage [pixel][channel] tiles in LDS by LDS-DMA and read them back with ds_read_b64_tr_b16.
//   wgrad_taps.hip  dense 3x3 / stride-1 convs: one block owns all nine taps of a (cout, cin) tile
//   wgrad_s2d.hip   3x3 / stride-2 convs over the space-to-depth copy of their input: all nine taps, four phase sub-tiles
//   wgrad_dma.hip   every other shape: one block per (tap, cout tile, cin tile)
// Partial sums over the pixel splits are combined with fp32 atomics (+= semantics).  Roofline: MFMA-bound,
// flops = 2*M*cout*ntaps*cin per launch.
#include "common.h"

using namespace nbdt;

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
  NBDT_REQUIRE(d->cu_budget == 0 || (d->cu_budget >= 32 && d->cu_budget <= 256), "cu_budget must be 0 or 32..256");
  const int64_t M_all = (int64_t)d->B * d->gh * d->gw;
  NBDT_REQUIRE(M_all < (1ll << 31), "pixel grid too large");
  if (nbdt::wgrad_taps_applicable(d)) return nbdt::wgrad_taps(d, x, gy, dw, (hipStream_t)stream);
  // 3x3 stride-2 over the space-to-depth copy of the input (round 6; variant 3 = the first-generation kernel, A/B and tests)
  if (d->variant != 3 && nbdt::wgrad_s2d_applicable(d)) return nbdt::wgrad_s2d(d, x, gy, dw, (hipStream_t)stream);
  NBDT_REQUIRE(d->variant == 0 || d->variant == 3, "variant 2 / 4 / 5 select between the dense 3x3 stride-1 kernels only");
  nbdt::g_last_wgrad = "conv_wgrad_dma_kernel";
  return nbdt::wgrad_dma(d, x, gy, dw, (hipStream_t)stream);
}

extern "C" int nbdt_conv_wgrad_blocks(const nbdt_wgrad_desc* d) {
  if (!d || d->cin <= 0 || d->cin % 32 != 0 || d->cout <= 0 || d->cout % 32 != 0 || d->ntaps != 9) return 0;
  if (d->B <= 0 || d->gh <= 0 || d->gw <= 0) return 0;
  if (d->cu_budget != 0 && (d->cu_budget < 32 || d->cu_budget > 256)) return 0;
  if (d->variant != 3 && nbdt::wgrad_s2d_applicable(d)) return nbdt::wgrad_s2d_blocks(d);
  return nbdt::wgrad_taps_blocks(d);
}

namespace nbdt { thread_local const char* g_last_wgrad = ""; }
extern "C" const char* nbdt_debug_last_wgrad(void) { return nbdt::g_last_wgrad; }
