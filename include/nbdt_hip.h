/*
 * nbdt_hip.h -- C-ABI of libnbdt_hip.so, the MI355X (gfx950) native hot path of
 * Neural-Backed Decision Trees.
 *
 * The reference (alvinwan/neural-backed-decision-trees) is 100% Python and has no
 * FFI/plugin registry: its boundary for this path is the nn.Module API
 *   nbdt/model.py:65-273   EmbeddedDecisionRules / Hard* / Soft*   (rules layer)
 *   nbdt/loss.py:97-266    TreeSupLoss / SoftTreeSupLoss           (tree-supervision loss)
 *   nbdt/models/resnet.py:42-149, nbdt/models/wideresnet.py:1-40   (backbones; every op a
 *                          stock aten/cuDNN kernel: Conv2d, BatchNorm2d, ReLU, avg-pool, Linear)
 *   main.py:207,233-239    SGD(momentum .9, wd 5e-4) train step
 * Each entry point below replaces the stock-op sequence named in its comment.  The host side
 * (neural-backed-decision-trees_amd/nbdt, Python, same class names/signatures as the
 * reference) binds these symbols with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer borrowed for the duration
 *     of the stream-ordered call unless marked "host"; the caller (PyTorch caching allocator)
 *     owns all tensors.  The library owns only tree handles.
 *   - every call returns 0 on success, a negative NBDT_E* code on failure and never throws;
 *     nbdt_last_error() returns a thread-local message for the last failure.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous and re-entrant across streams/devices.
 *   - activations are "padded NHWC bf16": [B][H+2][W+2][C] with a zero one-pixel border
 *     (so every 3x3 tap is in-bounds); kernels write interiors only.
 *   - conv weights are [Cout][taps][Cin] bf16 ("KRSC"), fp32 masters in the same order.
 */
#ifndef NBDT_HIP_H
#define NBDT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBDT_OK 0
#define NBDT_EINVAL (-1)   /* bad argument / unsupported shape */
#define NBDT_EHIP (-2)     /* HIP runtime error */
#define NBDT_ENOMEM (-3)

/* element types of the logits handed to the rules layer */
#define NBDT_F32 0
#define NBDT_BF16 1
#define NBDT_F16 2

const char* nbdt_last_error(void);
/* Name of the device kernel the calling thread's last nbdt_conv_igemm* call launched ("conv3x3_pp_kernel",
 * "conv3x3_halo_kernel", "conv_igemm_dma_kernel"): lets the parity tests assert WHICH kernel they exercised. */
const char* nbdt_debug_last_igemm(void);
/* the same with the kernel's template arguments, spelled as rocprofv3 prints it ("conv3x3_pp_kernel<5, false, 0, 8, false,
 * 2>"): what bench.py compares with the kernel names of a committed PMC traffic file before quoting it */
const char* nbdt_debug_last_igemm_full(void);
const char* nbdt_debug_last_wgrad(void);      /* same for nbdt_conv_wgrad */
int nbdt_version(void);
/* Deterministic mode (process-wide switch, default off).  The reference's CPU path (stock ATen ops,
 * nbdt/models/resnet.py:69-74) gives the same bits on every run; the fast path does not: every cross-block fp32
 * reduction of the backbone -- the 32 replicated BatchNorm accumulators (nbdt_bn_stats, nbdt_bn_bwd_reduce[_cus],
 * nbdt_pool_bn_bwd_reduce), the split-K weight gradients (nbdt_conv_wgrad, nbdt_stem_wgrad, nbdt_linear_bwd) and the
 * per-tile statistics of the conv epilogues (nbdt_conv_igemm_stats, nbdt_conv_igemm_bnbwd) -- goes through fp32
 * atomics whose order changes from run to run, and a 1-ulp difference in a BatchNorm sum moves bf16 roundings and
 * ReLU masks downstream (two runs of one WRN-28-10 step differ by ~0.17 relative L2 in the gradient).  With the switch
 * on, each of those reductions adds into a zeroed, library-owned row per block / per pixel split (one add per address)
 * and a fold kernel sums the rows in index order: two runs of the same launches on the same inputs are then
 * bit-identical, whatever streams they were issued on.  Slower (an extra fold per reduction, 64+ MB of workspace per
 * (device, stream), allocated with hipMalloc on first use -- not capturable in a hipGraph); the arithmetic differs
 * from the default mode only in summation order.  Scope: every kernel of the three backbones -- the EfficientNet-
 * specific reductions too (nbdt_dwconv_fwd's statistics, nbdt_dwconv_bwd_weight, nbdt_bn_act_pool, nbdt_bn_act_bwd,
 * nbdt_se_gate_bwd's parameter gradients) -- and the rules layer, which has no atomics. */
int nbdt_set_deterministic(int32_t on);
int nbdt_get_deterministic(void);
/* Epilogue of the K-split weight-gradient kernel (nbdt_conv_wgrad on the 3x3 stride-1 shapes; process-wide, default 1).
 * 1: each (pixel split, tile) block writes its partial sums with plain stores into that split's copy of dw in the
 *    per-(device, stream) workspace (the one deterministic mode uses, same hipGraph rule: one eager launch first) and
 *    a streaming pass adds the copies to dw in split order: the sum no longer depends on block timing.  The L2 retires
 *    fp32 atomics at ~1.2 TB/s whatever their shape: alone the launch is 4-8 % faster on the WRN-28-10 shapes and 20-27 %
 *    on ResNet18's at batch 128 (fold included); training steps: WRN-28-10 unchanged, ResNet18 1-4 % faster
 *    (profiles/r05_wgrad_store_epilogue_ab.txt).  Launches with more than 64 pixel splits keep the atomics, and so do
 *    CU-budgeted launches (desc.cu_budget > 0: an HBM-bound pass runs beside them, the fold would compete with it).
 * 0: fp32 atomics into dw (rounds 3-4).  Deterministic mode always takes the stores for that kernel. */
int nbdt_set_wgrad_store_epilogue(int32_t on);
int nbdt_get_wgrad_store_epilogue(void);
/* CUs (0..128, process-wide, default 0) the one-block-per-CU MFMA kernels leave free: the persistent forward / data-
 * gradient kernel launches 8 x (32 - ceil(n / 8)) blocks instead of 256 and the weight gradient is sized for at most
 * 256 - n.  For data-parallel training: RCCL's all-reduce kernels (one block per channel, NCCL_MAX_NCHANNELS of them)
 * run beside the backward pass, and a persistent block that finds its CU held by one of them starts when that block
 * ends -- the whole launch then waits for it (measured with a look-alike holder kernel: up to +1.7 ms per step).  The
 * engine sets n = the channel count while gradient buckets are in flight and 0 otherwise.  Results do not change
 * (same tiles, other block -> tile assignment).  Replaces nothing in the reference (DataParallel, main.py:160-162). */
int nbdt_set_reserved_cus(int32_t n);
int nbdt_get_reserved_cus(void);
/* number of visible HIP devices (0 => the product path must refuse to run) */
int nbdt_device_count(void);

/* DMA-ordered weight tiles for the dense 3x3 kernel.  For every listed matrix W[rows][9][k] (bf16, rows % 32 == 0,
 * k % 32 == 0; the forward weights, or the transposed tap-reversed data-gradient copy) write tiles
 * [row block of 32*nt][k slice of 32][tap] of (32*nt) x 32 elements, each stored exactly as the kernel's LDS image
 * (row-major 64-byte rows with the 16-byte chunk index XOR (row>>2)&3), so that one 1-KiB LDS-DMA instruction reads
 * one contiguous KiB instead of sixteen 64-byte fragments 2.8 KB apart.  nt = 5 | 4 | 2 | 1 is the largest of these
 * dividing rows/32 (the kernel's cout tile).  table (device, int64 [n][5]): {src element offset, dst element offset,
 * rows, k, first tile index}; total_tiles = sum of (rows/(32*nt)) * (k/32) * 9. */
int nbdt_weight_tile_batched(const void* src_bf16, const int64_t* table, int32_t n, int64_t total_tiles,
                             void* dst_bf16, void* stream);

/* ------------------------------------------------------------------ hierarchy handle
 * Flattened form of nbdt/tree.py Tree/Node (build_class_mappings :105-125):
 *   N inner nodes in `tree.inodes` order (sorted wnid), node n owns child slots
 *   [node_off[n], node_off[n+1]);  slot s averages the logits of classes
 *   slot_cls[slot_off[s] .. slot_off[s+1]) (ascending);  class c lies under slots
 *   cls_slot[cls_off[c] .. cls_off[c+1]) (inode order);  slot_next[s] = inode index of the
 *   child if it is an inner node, else -(class_index)-1.   All arrays are HOST pointers and are
 *   copied to `device`, together with the order in which a sample's lanes take the slots (longest first, dealt to
 *   the waves by load: csrc/rules.hip build_slot_schedule).
 * Size limits of the rules kernels (NBDT_EINVAL "hierarchy too large for LDS" beyond them): the per-sample rows
 *   (2-3 floats per class and per child slot) plus the offset arrays must fit 160 KB of LDS; when the sum of the
 *   leaf depths L = slot_off[R] also fits, the kernels stage the gathered operands there (ImageNet-1000: L = 11012),
 *   otherwise they index through the maps directly -- same results, slower. */
typedef struct nbdt_tree nbdt_tree;
int nbdt_tree_create(int device, int num_classes, int num_inodes, int root,
                     const int32_t* node_off, const int32_t* slot_off, const int32_t* slot_cls,
                     const int32_t* cls_off, const int32_t* cls_slot, const int32_t* slot_next,
                     nbdt_tree** out);
int nbdt_tree_destroy(nbdt_tree* t);
int nbdt_tree_max_depth(const nbdt_tree* t);

/* ------------------------------------------------------------------ rules layer
 * z: [B, C] logits, row stride ldz elements, element type ztype.  Outputs are fp32/int64,
 * dense row-major. */

/* SoftEmbeddedDecisionRules.forward (nbdt/model.py:207-242, 268-273): P[B,C] path probabilities */
int nbdt_soft_forward(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                      float* P, void* stream);
/* autograd of the above: gz[B,C] (fp32) = d<gP,P>/dz ; recomputes the forward from z */
int nbdt_soft_backward(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                       const float* gP, float* gz, void* stream);
/* SoftTreeSupLoss.forward + backward fused (nbdt/loss.py:191-203, 260-266) for
 * criterion = nn.CrossEntropyLoss():  loss = w_x*CE(z,y) + w_t*CE(P,y) (mean over B),
 * gz = grad_scale * dloss/dz.  row_loss: [B] fp32 scratch; loss: 1 fp32. */
int nbdt_soft_tree_loss(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                        const int64_t* y, float w_xent, float w_tree, float grad_scale,
                        float* row_loss, float* loss, float* gz, void* stream);
/* N1 -- classifier head + SoftTreeSupLoss forward AND backward in one launch: the logits never touch HBM.
 * Replaces nn.Linear (reference nbdt/models/resnet.py:126,148; pytorchcv `output`), TreeSupLoss.forward with the soft
 * rules (nbdt/loss.py:191-203, 264-266; nbdt/model.py:94-99, 207-242) and the Linear's autograd, i.e. the sequence
 * nbdt_linear_fwd -> nbdt_soft_tree_loss -> nbdt_linear_bwd, for classifiers of at most 512 classes / child slots
 * (NBDT_EINVAL beyond: use that sequence).
 *   pooled [B][K] fp32 features, W [C][K], bias [C] (nullable) fp32;  z = pooled W^T + bias (one wave per class,
 *   lanes over K, fused multiply-adds + xor butterfly: bit-identical to nbdt_linear_fwd for C < 64);
 *   loss / row_loss as nbdt_soft_tree_loss;  z_out [B][C] (nullable) = the logits the loss was computed on;
 *   gpooled [B][K] (nullable) = dL/dpooled;  gW [C][K], gb [C] (nullable) ACCUMULATE dL/dW, dL/db (+=). */
int nbdt_head_soft_tree_loss(const nbdt_tree* t, const float* pooled, const float* W, const float* bias, int64_t B,
                             int32_t K, const int64_t* y, float w_xent, float w_tree, float grad_scale,
                             float* row_loss, float* loss, float* z_out, float* gpooled, float* gW, float* gb,
                             void* stream);

/* HardTreeSupLoss.forward + backward fused (nbdt/loss.py:191-203, 212-257; label filtering
 * nbdt/model.py:127-143) for criterion = nn.CrossEntropyLoss():
 *   loss = mean_b[ w_xent*CE(z_b,y_b) + w_node * sum_{inner nodes n with y_b under n}
 *                                                   CE(node_logits(z_b, n), child_of(n, y_b)) ]
 * the caller folds the reference's weights into w_node = tree_weight * tsw * 2 / N (every pooled
 * (sample,node) row weighs tsw/(B*N/2), then TreeSupLoss.forward applies the schedule).
 * gz = grad_scale * dloss/dz.  row_loss: [B] fp32 scratch; loss: 1 fp32. */
int nbdt_hard_tree_loss(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                        const int64_t* y, float w_xent, float w_node, float grad_scale,
                        float* row_loss, float* loss, float* gz, void* stream);
/* VJP of get_node_logits over every inner node (nbdt/model.py:83-99): gs [B,R] fp32 gradient of
 * the child logits (slot-major, as written by nbdt_node_outputs) -> gz [B,C] fp32. */
int nbdt_node_logits_backward(const nbdt_tree* t, const float* gs, int64_t B, float* gz, void* stream);
/* HardEmbeddedDecisionRules.forward_with_decisions (nbdt/model.py:145-199): pred[B] int64,
 * optional onehot[B,C] fp32 (predicted_to_logits), optional decision buffers
 * [B, max_depth]: inode index / chosen child / its prob / node entropy (-1 padded). */
int nbdt_hard_forward(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                      int64_t* pred, float* onehot, int32_t* path_node, int32_t* path_child,
                      float* path_prob, float* path_entropy, void* stream);
/* EmbeddedDecisionRules.forward_nodes (nbdt/model.py:101-123): per-slot logits/probs [B,R],
 * per-node preds [B,N] int64 and entropy [B,N]; any output may be NULL. */
int nbdt_node_outputs(const nbdt_tree* t, const void* z, int ztype, int64_t B, int64_t ldz,
                      float* logits, float* probs, int64_t* preds, float* entropy, void* stream);

/* ------------------------------------------------------------------ backbone: implicit-GEMM conv
 * One launch = one "tap table": out[pix(m)][n] (+)= sum_t sum_c in[pix_in(m) + tap_off[t]][c] *
 * w[n][w_tap[t]][c].  Pixel m of the launch's M-pixel grid decomposes as (b, i, j) over
 * (B, gh, gw); element offsets are affine: b*bs + i*hs + j*ws + base.  This single kernel is
 * Conv2d forward (3x3/1x1, stride 1/2; nbdt/models/resnet.py:47-66), its dgrad (flipped taps,
 * parity classes for stride 2) and the 1x1 shortcut, replacing cuDNN/MIOpen.           */
typedef struct nbdt_conv_desc {
  int32_t B, gh, gw;            /* pixel grid of this launch: M = B*gh*gw */
  int32_t cin, cout;            /* cin % 32 == 0, cout % 32 == 0 */
  int32_t ntaps;                /* <= 9 */
  int32_t tap_off[9];           /* element offset added to the input pixel offset */
  int32_t w_tap[9];             /* tap index into the weight tensor */
  int32_t w_ntaps;              /* taps stored in the weight tensor (row length = w_ntaps*cin) */
  int32_t in_bs, in_hs, in_ws, in_base;      /* element strides of the input pixel map */
  int32_t out_bs, out_hs, out_ws, out_base;  /* element strides of the output pixel map */
  int32_t accumulate;           /* 1: out += result (reads out) */
  int32_t wide_tile;            /* dense 3x3 stride-1 launches with w_tiled: 0 / 1 = pick the tile from the grid size (the
                                   8-wave ping-pong kernel on 512-pixel tiles when they give >= 3/4 of the CUs a block,
                                   on 256-pixel half tiles when those fit the CUs in one round, else 512-pixel tiles);
                                   2 = force 512-pixel tiles (error if the shape does not fit), 3 = force the 4-wave
                                   256-pixel kernel, 4 = force 512-pixel tiles with the padded LDS pitch (images narrower
                                   than 32 pixels: bank-conflict-free halo reads, measured 1-2 % slower, so never picked
                                   automatically), 5 = force half tiles.  2 - 5 exist for tests and A/B measurements. */
  int32_t ksplit;               /* (a reserved, ignored field before nbdt_version() 106: zero-initialise descriptors;
                                   negative values, and n > 1 without wide_tile = 5, are NBDT_EINVAL)
                                   half tiles on at most half the CUs: blocks per output tile, each a range of the 32-channel
                                   input slices (partials through a per-stream fp32 workspace, the last block sums them in
                                   split order and runs the epilogue).  0 = automatic (only with an automatic wide_tile),
                                   1 = never, n = n blocks per tile (tests, A/B; with wide_tile = 5) */
  uint64_t w_tiled;             /* 0, or device pointer to the same weights pre-arranged by nbdt_weight_tile_batched
                                   (only dense 3x3 stride-1 launches with the identity tap map use it) */
} nbdt_conv_desc;
/* Host only, no device work: which kernel form the launches below pick for this descriptor and this process's reserved
 * CUs (nbdt_set_reserved_cus) -- so that the launch rules can be tested and planned with without a GPU.
 * form: 0 = first-generation implicit GEMM (strided / 1x1 / anything that is not a dense 3x3 stride-1 conv over a padded
 * tensor), 1 = 4-wave 256-pixel kernels (no DMA-ordered weights), 2 = ping-pong kernel on 512-pixel tiles, 3 = the same with
 * the padded LDS pitch, 4 = ping-pong kernel on 256-pixel half tiles; ksplit: blocks per half tile the rule asks for
 * (form 4; a launch without its per-stream workspace -- first use inside a hipGraph capture -- runs unsplit). */
int nbdt_conv_plan(const nbdt_conv_desc* d, int32_t* form, int32_t* ksplit);
/* in/out/w bf16; residual (nullable) bf16 addressed like out and added before rounding */
int nbdt_conv_igemm(const nbdt_conv_desc* d, const void* in, const void* w, void* out,
                    const void* residual, void* stream);

/* Up to four nbdt_conv_igemm launches over the SAME operands in one grid (no residual, no statistics; `accumulate`
 * must agree): the output-parity classes of a strided 3x3 data gradient write disjoint pixels, and each alone fills a
 * quarter of the chip at 512 images.  Same arithmetic per descriptor as nbdt_conv_igemm (bit-identical outputs).
 * Replaces the conv-transpose autograd of a strided nn.Conv2d (pytorchcv PreResUnit's first conv of a stage). */
int nbdt_conv_igemm_multi(const nbdt_conv_desc* descs, int32_t n, const void* in, const void* w, void* out,
                          void* stream);

/* ------------------------------------------------------------------ backbone: "slice list" convolutions (round 6)
 * The shape-changing units of a (Wide)ResNet -- a stride-2 3x3 conv, its 1x1 stride-2 shortcut and their data
 * gradients (nbdt/models/resnet.py:56-67; pytorchcv PreResUnit with stride 2 / a channel change behind
 * nbdt/models/wideresnet.py:1-5) -- are not dense 3x3 stride-1 convolutions, but every one of them IS a sum of
 * stride-1 tap-subset convolutions over tensors that share one padded pixel grid:
 *   - the stride-2 forward conv over the space-to-depth copy of its input (nbdt_bn_apply_s2d): phase (p,q) of the
 *     input holds 4 / 2 / 2 / 1 of the nine taps;
 *   - the four output-parity classes of its data gradient over the output gradient (4 / 2 / 2 / 1 taps, strided
 *     stores), the shortcut's data gradient being one more 1-tap term of class (even, even);
 *   - conv2 + shortcut of the same unit: nine taps over conv2's input plus one tap over the unit's (space-to-depth)
 *     input -- the residual add disappears into the K loop.
 * A launch is therefore described as up to four CLASSES (disjoint output pixel maps over the same pixel grid), each an
 * ordered list of 32-channel K SLICES: (input tensor, first channel, tap subset, where its weights are).  One kernel
 * (csrc/conv_seg.hip: 8-wave ping-pong, LDS-resident halo slices, persistent blocks) runs them all.  The plan object
 * owns the device-side step tables (which LDS-DMA piece goes out in which K step); weights are re-tiled into DMA
 * order by nbdt_conv_seg_tile_weights whenever they change. */
typedef struct nbdt_conv_seg_slice {
  int32_t tensor;               /* input tensor index 0..3 */
  int32_t ch0;                  /* first of the slice's 32 channels inside a pixel of that tensor (multiple of 8) */
  int32_t ntaps;                /* 1..9 */
  int32_t tap[9];               /* 3*R + S: input pixel (y + R - 1, x + S - 1) of the launch's pixel grid */
  int32_t w_matrix;             /* weight matrix index 0..3 */
  int32_t w_off[9];             /* per tap: element offset, inside a row of that matrix, of the slice's 32 k-values */
} nbdt_conv_seg_slice;
typedef struct nbdt_conv_seg_class {
  int32_t nslices;              /* 1..NBDT_SEG_MAX_SLICES */
  const nbdt_conv_seg_slice* slices;
  int32_t out_bs, out_hs, out_ws, out_base;   /* element strides of the class's output pixel map (like nbdt_conv_desc) */
} nbdt_conv_seg_class;
#define NBDT_SEG_MAX_SLICES 64
typedef struct nbdt_conv_seg_desc {
  int32_t B, gh, gw;            /* pixel grid shared by every input tensor ([B][gh+2][gw+2][pix_stride], zero border)
                                   and by the classes' output maps */
  int32_t cout;                 /* multiple of 32; rows of every weight matrix */
  int32_t ntensors;             /* 1..4 */
  int32_t pix_stride[4];        /* elements per pixel of each input tensor */
  int32_t nmatrices;            /* 1..4 */
  int32_t w_row_stride[4];      /* elements per row of each weight matrix ([cout][row_stride] bf16) */
  int32_t nclasses;             /* 1..4 */
  nbdt_conv_seg_class cls[4];
  int32_t tile;                 /* 0 = pick (512-pixel tiles when they give >= 3/4 of the CUs a block, else 256), 512, 256 */
  int32_t nbuf;                 /* 0 = pick (2 halo buffers when every slice but a class's last has >= 2 taps, else 3), 2, 3 */
} nbdt_conv_seg_desc;
/* Host side only (no GPU needed until the first launch allocates the device tables): validates the description, picks
 * tile / buffers, schedules the LDS-DMA pieces of every slice over the K steps before it; NBDT_EINVAL when the shape
 * does not fit (tiles must be whole image rows / whole images, the halo buffers + weight ring must fit in 160 KB). */
int nbdt_conv_seg_create(const nbdt_conv_seg_desc* d, void** plan);
int nbdt_conv_seg_destroy(void* plan);
/* what the plan chose: tile pixels, halo buffers, K steps per class (steps[4]), max LDS-DMA rounds in one step, bf16
 * elements of the DMA-ordered weight buffer */
int nbdt_conv_seg_info(void* plan, int32_t* tile, int32_t* nbuf, int32_t* steps, int32_t* max_rounds,
                       int64_t* w_tile_elems);
/* w[nmatrices] bf16 device pointers -> w_tiles (w_tile_elems bf16): per class, per cout tile, per K step one
 * (32*NT rows) x 32 k tile stored as the swizzled LDS image it will be copied to */
int nbdt_conv_seg_tile_weights(void* plan, const void* const* w, void* w_tiles, void* stream);
/* in[ntensors] bf16 device pointers; residual (nullable) is addressed like class 0's output; bn_partials (nullable,
 * single-class launches only) as in nbdt_conv_igemm_stats */
int nbdt_conv_seg(void* plan, const void* const* in, const void* w_tiles, void* out, const void* residual,
                  float* bn_partials, void* stream);
/* host copy of one class's K-step records (8 int32 each: tap offset in halo pixels, halo buffer byte offset, LDS-DMA
 * rounds issued in the step, their LDS destination, first halo pixel, channel, tensor, strict flag) and the number of
 * slices its tiles load in their prologue -- so that the DMA schedule can be checked without a GPU.  Returns the
 * number of steps (>= 0) or an error (< 0). */
int nbdt_conv_seg_steps(void* plan, int32_t cls, int32_t* out8, int32_t max_steps, int32_t* npro);

/* same launch, and the epilogue also writes the per-channel sum / sum of squares of the (bf16) output
 * of every 256-pixel tile to bn_partials[ceil(M/256)][2][cout] (plain stores, fully overwritten) -- the
 * statistics the following BatchNorm needs, so nbdt_bn_finalize can replace nbdt_bn_stats (no second
 * pass over the tensor). */
int nbdt_conv_igemm_stats(const nbdt_conv_desc* d, const void* in, const void* w, void* out,
                          const void* residual, float* bn_partials, void* stream);

/* data-gradient launch whose output is dL/d(relu(bn(x))): the epilogue also reads x (the BatchNorm's
 * forward input, same geometry as `out`) and writes per-tile partial sums of g' and g'*xhat, g' = out *
 * [bn(x) > 0] -- the reductions of the BatchNorm backward (nbdt_bn_bwd_fold folds them; replaces
 * nbdt_bn_bwd_reduce and one full re-read of the gradient tensor). */
int nbdt_conv_igemm_bnbwd(const nbdt_conv_desc* d, const void* in, const void* w, void* out,
                          const void* bn_x, const float* save_mean, const float* save_rstd,
                          const float* gamma, const float* beta, float* bn_partials, void* stream);
/* inference: eval-mode BatchNorm folded into per-channel scale/shift (scale = gamma/sqrt(running_var+eps),
 * shift = beta - running_mean*scale) and the activation applied in the conv epilogue:
 *   out = act(conv(in) * scale[c] + shift[c] [+ residual]),  act: NBDT_ACT_NONE | RELU | SWISH.
 * Replaces Conv2d -> BatchNorm2d(eval) -> ReLU [-> += shortcut] of nbdt/models/resnet.py:69-74 in ONE launch. */
int nbdt_conv_igemm_affine(const nbdt_conv_desc* d, const void* in, const void* w, void* out,
                           const void* residual, const float* scale, const float* shift, int32_t act,
                           void* stream);

/* weight gradient (replaces cuDNN wgrad): dw[cout][w_ntaps][cin] fp32 += sum over the pixel grid
 * of gy[pix_g(m)][co] * x[pix_x(m) + tap_off[t]][ci]; split over pixels with fp32 atomics, so dw
 * must be zeroed (or hold the running .grad) before the call. */
typedef struct nbdt_wgrad_desc {
  int32_t B, gh, gw;
  int32_t cin, cout;
  int32_t ntaps;
  int32_t tap_off[9];
  int32_t w_tap[9];
  int32_t w_ntaps;
  int32_t x_bs, x_hs, x_ws, x_base;
  int32_t g_bs, g_hs, g_ws, g_base;
  int32_t variant;              /* dense 3x3 stride-1 launches: 0 = pick from the problem size (the K-split 8-wave kernel
                                   from 64 pixel stages on, else the 4-wave one); 2 = force the tap-split 8-wave kernel
                                   (two wave groups, taps divided between them), 3 = force the 4-wave one, 4 = force the
                                   12-wave kernel (three wave groups, one kernel row each), 5 = force the K-split 8-wave
                                   kernel (every wave all nine taps, the groups halve the pixels) -- tests, A/B */
  int32_t cu_budget;            /* dense 3x3 stride-1 launches: 0 = size the pixel split for all 256 CUs; n = for n of
                                   them (32..256), so that an HBM-bound pass launched on another stream keeps the
                                   rest: a weight-gradient block takes a CU's whole register file, the two kernels
                                   never share one (probes/cu_share_probe.hip) */
} nbdt_wgrad_desc;
int nbdt_conv_wgrad(const nbdt_wgrad_desc* d, const void* x, const void* gy, float* dw,
                    void* stream);
/* thread blocks the launch above will use for this descriptor (its cu_budget included) when it takes the 8-wave dense
 * 3x3 kernel -- one block per CU, so 256 minus this is what a concurrent HBM-bound pass may take
 * (nbdt_bn_bwd_apply_cus); 0 for every other kernel.  No device work. */
int nbdt_conv_wgrad_blocks(const nbdt_wgrad_desc* d);

/* fp32 master [cout][taps][cin] -> bf16 copy in the same order, and (optional) the dgrad copy
 * wd[cin][taps][cout] with the tap order reversed (wd[ci][t][co] = w[co][taps-1-t][ci]) */
int nbdt_weight_prep(const float* w, int32_t cout, int32_t taps, int32_t cin, void* w_bf16,
                     void* wd_bf16, void* stream);

/* the dgrad copies of EVERY conv layer in one launch: table (device, int64 [n_layers][6]) rows are
 * {src offset in flat, dst offset in wd_flat, cout, taps, cin, first tile index}; cout, cin % 32 == 0; a tile
 * is 64 couts x 32 cins of one tap: total_tiles = sum over layers of taps * ceil(cout/64) * cin/32 */
int nbdt_weight_prep_batched(const float* flat, const int64_t* table, int32_t n_layers, int64_t total_tiles,
                             void* wd_flat, void* stream);

/* ------------------------------------------------------------------ backbone: batch-norm / elementwise
 * Replaces nn.BatchNorm2d (train mode, eps 1e-5, momentum 0.1) + F.relu + residual adds
 * (nbdt/models/resnet.py:69-74; pytorchcv PreResUnit) and their autograd.  Tensors are padded
 * NHWC bf16 [B][H+2][W+2][C], C % 8 == 0; only interiors are read/written.  `scratch` is
 * NBDT_BN_SLOTS*2*C fp32 of caller-owned workspace that must be ZERO on entry; every call leaves it
 * zero again (the fold kernel clears what it read), so one zero-initialised buffer serves all layers. */
#define NBDT_BN_SLOTS 32
/* batch statistics: save_mean/save_rstd [C]; updates running_mean/var (unbiased var) if non-NULL.
 * x == NULL: only fold sums a producer already left in `scratch` (nbdt_dwconv_fwd with bn_scratch). */
int nbdt_bn_stats(const void* x, int32_t B, int32_t H, int32_t W, int32_t C, float eps,
                  float momentum, float* running_mean, float* running_var, float* scratch,
                  float* save_mean, float* save_rstd, void* stream);
/* fold the partial sums nbdt_conv_igemm_stats wrote for a [B,H,W,C] output (ceil(B*H*W/256) rows) into
 * save_mean/save_rstd + running statistics */
int nbdt_bn_finalize(int32_t B, int32_t H, int32_t W, int32_t C, float eps, float momentum,
                     float* running_mean, float* running_var, const float* bn_partials, float* save_mean,
                     float* save_rstd, void* stream);
/* y = relu?( (x-mean)*rstd*gamma + beta [+ residual] ) */
int nbdt_bn_apply(const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                  const float* beta, const void* residual, int32_t relu, int32_t B, int32_t H,
                  int32_t W, int32_t C, void* y, void* stream);
/* the same pass writing the SPACE-TO-DEPTH copy of its output: y is [B][H/2+2][W/2+2][4C] bf16 (zero border), input
 * pixel (h, w) at pixel (h/2, w/2), channels [((h&1)*2 + (w&1))*C, +C).  What the stride-2 conv1 and the 1x1 stride-2
 * shortcut of a shape-changing unit (and their weight gradients) read instead of the plain activated tensor:
 * pytorchcv PreResUnit's shared pre-activation (nbdt/models/wideresnet.py:1-5), nbdt/models/resnet.py:56-67. */
int nbdt_bn_apply_s2d(const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                      const float* beta, int32_t relu, int32_t B, int32_t H, int32_t W, int32_t C, void* y,
                      void* stream);
/* backward, pass 1: with gy' = gy * mask when relu, writes dsum[0][c] = sum gy',
 * dsum[1][c] = sum gy' * xhat and accumulates dbeta += dsum[0], dgamma += dsum[1] (either may be
 * NULL).  mask = (y > 0) from the stored forward output y; with y == NULL (no residual in the
 * forward) it is recomputed from x, gamma, beta with bn_apply's own expression -- one tensor read less. */
int nbdt_bn_bwd_reduce(const void* gy, const void* y, const void* x, const float* save_mean,
                       const float* save_rstd, const float* gamma, const float* beta, int32_t relu,
                       int32_t B, int32_t H, int32_t W, int32_t C, float* scratch, float* dsum,
                       float* dgamma, float* dbeta, void* stream);
/* fold the partials of nbdt_conv_igemm_bnbwd into dsum[2][C]; dbeta += dsum[0], dgamma += dsum[1] */
int nbdt_bn_bwd_fold(int32_t B, int32_t H, int32_t W, int32_t C, const float* bn_partials, float* dsum,
                     float* dgamma, float* dbeta, void* stream);
/* backward, pass 2: gx = gamma*rstd*(gy' - (dsum0 + xhat*dsum1)/N) [+ gx_add];
 * g_resid (nullable) receives gy' (the gradient of the residual input). */
int nbdt_bn_bwd_apply(const void* gy, const void* y, const void* x, const float* save_mean,
                      const float* save_rstd, const float* gamma, const float* beta, const float* dsum,
                      const void* gx_add, int32_t relu, int32_t B, int32_t H, int32_t W, int32_t C,
                      void* gx, void* g_resid, void* stream);
/* nbdt_bn_bwd_apply with relu = 1, y = NULL, g_resid = NULL on `cus` CUs only (one persistent block each): the pass is
 * HBM-bound and needs few of them (64 CUs stream 3.0 TB/s, 96 4.1 TB/s, all 256 5.6 TB/s), so an MFMA-bound launch on
 * another stream -- nbdt_conv_wgrad with nbdt_wgrad_desc.cu_budget = 256 - cus -- runs on the rest at the same
 * time.  (Blocks of the ordinary launch land on every CU and keep a weight-gradient block, which takes a CU's whole
 * register file, from starting.)  Same arithmetic, same results as nbdt_bn_bwd_apply. */
int nbdt_bn_bwd_apply_cus(const void* gy, const void* x, const float* save_mean, const float* save_rstd,
                          const float* gamma, const float* beta, const float* dsum, const void* gx_add,
                          int32_t B, int32_t H, int32_t W, int32_t C, void* gx, int32_t cus, void* stream);
/* nbdt_bn_bwd_reduce with relu = 1, y = NULL on `cus` CUs only (see nbdt_bn_bwd_apply_cus): with it the sums need not
 * come out of the data gradient's epilogue (nbdt_conv_igemm_bnbwd), and the whole BatchNorm backward -- reduce, fold,
 * apply -- is HBM-bound work that runs beside the weight gradient.  scratch: the zeroed 32-slot buffer of
 * nbdt_bn_bwd_reduce (left zeroed). */
int nbdt_bn_bwd_reduce_cus(const void* gy, const void* x, const float* save_mean, const float* save_rstd,
                           const float* gamma, const float* beta, int32_t B, int32_t H, int32_t W, int32_t C,
                           float* scratch, float* dsum, float* dgamma, float* dbeta, int32_t cus, void* stream);
/* nbdt_bn_bwd_reduce_cus + nbdt_bn_bwd_apply_cus as TWO launches instead of three: the fold of the 32 slots
 * (bn_bwd_finalize, 7 us on the backward critical path of every conv) happens in the prologue of the elementwise pass --
 * every block folds the slots for itself, block 0 writes dsum / accumulates dgamma, dbeta.  The slots being read cannot
 * be re-zeroed inside that launch, so the caller passes a PAIR of 32-slot buffers and alternates them from call to
 * call: `slots` (zero on entry, dirty on return) receives this call's sums, `slots_other` (not touched by anyone
 * while this call runs) is left zeroed for the next call.  Same arithmetic and summation order as the three-launch
 * form: bit-identical gx / dsum in deterministic mode.  Replaces the autograd of F.relu(bn(x)) (pytorchcv
 * PreResActivation behind nbdt/models/wideresnet.py:1-5). */
int nbdt_bn_bwd_cus(const void* gy, const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                    const float* beta, const void* gx_add, int32_t B, int32_t H, int32_t W, int32_t C, float* slots,
                    float* slots_other, float* dsum, float* dgamma, float* dbeta, void* gx, int32_t cus,
                    void* stream);
/* head: pooled[b][c] = mean over (h,w) of relu(bn(x))  (post_activ + final_pool / avg_pool2d,
 * nbdt/models/resnet.py:142) and its backward given gpooled[B][C] (same two passes) */
int nbdt_bn_relu_pool(const void* x, const float* save_mean, const float* save_rstd,
                      const float* gamma, const float* beta, int32_t B, int32_t H, int32_t W,
                      int32_t C, float* pooled, void* stream);
int nbdt_pool_bn_bwd_reduce(const float* gpooled, const void* x, const float* save_mean,
                            const float* save_rstd, const float* gamma, const float* beta,
                            int32_t B, int32_t H, int32_t W, int32_t C, float* scratch, float* dsum,
                            float* dgamma, float* dbeta, void* stream);
int nbdt_pool_bn_bwd_apply(const float* gpooled, const void* x, const float* save_mean,
                           const float* save_rstd, const float* gamma, const float* beta,
                           const float* dsum, int32_t B, int32_t H, int32_t W, int32_t C, void* gx,
                           void* stream);

/* ------------------------------------------------------------------ backbone: MBConv pieces (EfficientNet-B0)
 * Replaces pytorchcv `efficientnet_b0` building blocks behind nbdt/models/__init__.py:3:
 * dwconv{3x3,5x5}_block, BatchNorm2d + Swish, SEBlock, Dropout -- and their autograd.  Same padded NHWC
 * bf16 tensors as above.  The activated tensor a = act(bn(x)) is never stored for the SE branch: every
 * pass recomputes it from the raw conv output x and the saved batch statistics. */
#define NBDT_ACT_NONE 0
#define NBDT_ACT_RELU 1
#define NBDT_ACT_SWISH 2
/* y = act(bn(x)) [* gate[b][c]] [+ residual]        gate: fp32 [B][C] or NULL */
int nbdt_bn_act_apply(const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                      const float* beta, int32_t act, const float* gate, const void* residual, int32_t B,
                      int32_t H, int32_t W, int32_t C, void* y, void* stream);
/* out[b][c] = scale * sum_hw act(bn(x)) [* mul]      (SE squeeze / head average pool with scale = 1/HW;
 * with mul = upstream gradient and scale = 1: dL/dgate of the SE scaling).  out: fp32 [B][C], overwritten */
int nbdt_bn_act_pool(const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                     const float* beta, int32_t act, const void* mul, float scale, int32_t B, int32_t H,
                     int32_t W, int32_t C, float* out, void* stream);
/* backward of u = act(bn(x)) * gate (+ the pooled branch) w.r.t. x, gamma, beta in two passes:
 *   gradient entering the activation g_a = gu                               (gate == NULL, gu != NULL)
 *                                        = gu*gate[b][c] + gpool[b][c]/HW   (SE form)
 *                                        = gpool[b][c]/HW                    (gu == NULL: average-pool head)
 * dgamma/dbeta are accumulated (+=); gx [+= gx_add]; scratch/dsum as in nbdt_bn_bwd_reduce. */
int nbdt_bn_act_bwd(const void* gu, const float* gate, const float* gpool, const void* x,
                    const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                    int32_t act, const void* gx_add, int32_t B, int32_t H, int32_t W, int32_t C,
                    float* scratch, float* dsum, float* dgamma, float* dbeta, void* gx, void* stream);
/* depthwise Conv2d(C, C, k in {3,5}, stride in {1,2}, padding k/2, groups=C): x [B][H+2][W+2][C] ->
 * y [B][H/stride+2][W/stride+2][C]; w fp32 [k*k][C] (tap-major); dw accumulated (+=). */
int nbdt_dwconv_fwd(const void* x, const float* w, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                    int32_t stride, void* y, float* bn_scratch /* nullable: also accumulate sum(y), sum(y^2)
                    into the NBDT_BN_SLOTS scratch for nbdt_bn_stats(x = NULL, ...) */, void* stream);
int nbdt_dwconv_bwd_data(const void* gy, const float* w, int32_t B, int32_t H, int32_t W, int32_t C,
                         int32_t k, int32_t stride, void* gx, void* stream);
/* Stride-1 depthwise data gradient whose epilogue also produces the backward sums of the BatchNorm + swish that fed the
 * depthwise conv (MBConv: dwconv(swish(bn1(e))) -- pytorchcv dwconv block after the expand conv): gx = dL/d(swish(bn(bn_x))),
 * and sum(g'), sum(g' * xhat) with g' = gx * swish'(bn(bn_x)) go to the 32-slot `scratch` -- follow with
 * nbdt_bn_act_bwd_apply (fold + elementwise pass), which replaces nbdt_bn_act_bwd's reduction pass over gx and bn_x. */
int nbdt_dwconv_bwd_data_bn(const void* gy, const float* w, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                            void* gx, const void* bn_x, const float* save_mean, const float* save_rstd,
                            const float* gamma, const float* beta, float* scratch, void* stream);
/* nbdt_bn_act_bwd without its reduction pass: the sums are already in `scratch` (plain form only: no gate, no pool). */
int nbdt_bn_act_bwd_apply(const void* gu, const void* x, const float* save_mean, const float* save_rstd,
                          const float* gamma, const float* beta, int32_t act, const void* gx_add, int32_t B, int32_t H,
                          int32_t W, int32_t C, float* scratch, float* dsum, float* dgamma, float* dbeta, void* gx,
                          void* stream);
/* Squeeze-and-excitation backward with ONE reduction pass over (gu, x) instead of two (nbdt_bn_act_pool(mul = gu) for
 * dL/dgate, then nbdt_bn_act_bwd's sums).  The gradient entering the activation, g_a = gu*gate[b][c] + gpool[b][c]/HW, is
 * linear in (gate, gpool) per image and channel, so the BatchNorm-backward sums follow from five per-(image, channel) sums
 * that do not need gpool: sums[5][B][C] fp32 += { sum gu*act(y), sum gu*act'(y), sum gu*act'(y)*xhat, sum act'(y),
 * sum act'(y)*xhat } over the image's pixels, y = bn(x).  `sums` must be ZERO on entry (allocate it zeroed once); sums[0]
 * is dL/dgate: feed it to nbdt_se_gate_bwd, then nbdt_bn_act_se_bwd_apply folds sums[1..4] with gate / gpool into dsum
 * (dgamma, dbeta accumulate), ZEROES `sums` again for the next use (no memset launch per call) and runs
 * nbdt_bn_act_bwd's elementwise pass.  Replaces the autograd of pytorchcv's SEBlock scaling + BatchNorm + swish
 * (reference: nbdt/models/__init__.py:3).  Not available in deterministic mode (atomics across pixel slices). */
int nbdt_bn_act_se_sums(const void* gu, const void* x, const float* save_mean, const float* save_rstd,
                        const float* gamma, const float* beta, int32_t act, int32_t B, int32_t H, int32_t W, int32_t C,
                        float* sums, void* stream);
int nbdt_bn_act_se_bwd_apply(const void* gu, const float* gate, const float* gpool, float* sums, const void* x,
                             const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                             int32_t act, int32_t B, int32_t H, int32_t W, int32_t C, float* dsum, float* dgamma,
                             float* dbeta, void* gx, void* stream);
int nbdt_dwconv_bwd_weight(const void* x, const void* gy, int32_t B, int32_t H, int32_t W, int32_t C,
                           int32_t k, int32_t stride, float* dw, void* stream);
/* SEBlock gate: gate[b][c] = sigmoid(W2 swish(W1 pooled[b] + b1) + b2) for c < C_real, 0 above.
 * pooled/gate rows have stride C; W1 [S][C_real], W2 [C_real][S]; pre1 [B][S] saved for backward. */
int nbdt_se_gate_fwd(const float* pooled, const float* w1, const float* b1, const float* w2,
                     const float* b2, int32_t B, int32_t C, int32_t C_real, int32_t S, float* pre1,
                     float* gate, void* stream);
/* dgate [B][C] -> gpool [B][C] (gradient of the pooled mean) and += dW1, db1, dW2, db2;
 * dpre2 [B][C_real], dpre1 [B][S]: workspaces.  dw1 = db1 = dw2 = db2 = NULL: the data part only -- the parameter
 * gradients (which feed nothing but the optimizer) are then the caller's to launch, e.g. on a second stream, with
 * nbdt_se_param_grad on the dpre2 / dpre1 this call left (nbdt_version() >= 108). */
int nbdt_se_gate_bwd(const float* dgate, const float* gate, const float* pre1, const float* pooled,
                     const float* w1, const float* w2, int32_t B, int32_t C, int32_t C_real, int32_t S,
                     float* dpre2, float* dpre1, float* gpool, float* dw1, float* db1, float* dw2,
                     float* db2, void* stream);
int nbdt_se_param_grad(const float* dpre2, const float* dpre1, const float* pre1, const float* pooled,
                       int32_t B, int32_t C, int32_t C_real, int32_t S, float* dw1, float* db1, float* dw2,
                       float* db2, void* stream);
/* nn.Dropout(p) on n fp32 values: mask[i] in {0,1} from a counter hash of (seed, i); y = x*mask/(1-p) */
int nbdt_dropout_fwd(const float* x, int64_t n, float p, uint32_t seed, uint8_t* mask, float* y,
                     void* stream);
int nbdt_dropout_bwd(const float* gy, int64_t n, float p, const uint8_t* mask, float* gx, void* stream);

/* ------------------------------------------------------------------ verification-only fp32-storage kernels
 * NOT the product path (csrc/ref_fp32.hip): the same operators as above on fp32 padded NHWC tensors, written to be
 * obviously right (one thread per output, plain loops, double accumulators), taking the SAME descriptors.  The engines'
 * fp32 reference mode (engine.set_reference_fp32) routes every launch here while keeping its launch order, streams,
 * events and buffer rotation, so that ONE test can show the whole training step -- in the shipped schedule -- agreeing
 * with the fp32 oracle to 1e-5 instead of the cosine ~0.9 that bf16 storage allows (tests/test_reference_fp32_gpu.py).
 * They replace, for verification, the same reference lines as their product twins (nbdt/models/resnet.py:47-74,
 * 115-149; pytorchcv PreResUnit). */
int nbdt_ref_conv(const nbdt_conv_desc* d, const float* in, const float* w, float* out, const float* residual,
                  void* stream);
int nbdt_ref_wgrad(const nbdt_wgrad_desc* d, const float* x, const float* gy, float* dw, void* stream);
/* twin of nbdt_conv_seg: fp32 tensors, fp32 weight matrices in their plain [cout][row_stride] layout (no tiling) */
int nbdt_ref_conv_seg(void* plan, const float* const* in, const float* const* w, float* out, const float* residual,
                      void* stream);
/* partials == NULL: nbdt_bn_stats.  partials != NULL: the conv-epilogue form -- row 0 of bn_partials[rows][2][C]
 * receives sum / sum of squares, the other rows zero (nbdt_bn_finalize folds them as usual). */
int nbdt_ref_bn_stats(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float eps, float momentum,
                      float* running_mean, float* running_var, float* save_mean, float* save_rstd, float* partials,
                      void* stream);
int nbdt_ref_bn_apply(const float* x, const float* save_mean, const float* save_rstd, const float* gamma,
                      const float* beta, const float* residual, int32_t relu, int32_t B, int32_t H, int32_t W,
                      int32_t C, float* y, void* stream);
int nbdt_ref_bn_apply_s2d(const float* x, const float* save_mean, const float* save_rstd, const float* gamma,
                          const float* beta, int32_t relu, int32_t B, int32_t H, int32_t W, int32_t C, float* y,
                          void* stream);
/* nbdt_bn_bwd_reduce + nbdt_bn_bwd_apply (gy given) or nbdt_pool_bn_bwd_reduce + _apply (gy NULL, gpooled given);
 * reduce == 0: the elementwise pass only, with the caller's dsum */
int nbdt_ref_bn_bwd(const float* gy, const float* gpooled, const float* y, const float* x, const float* save_mean,
                    const float* save_rstd, const float* gamma, const float* beta, int32_t relu, const float* gx_add,
                    int32_t B, int32_t H, int32_t W, int32_t C, int32_t reduce, float* dsum, float* dgamma,
                    float* dbeta, float* gx, float* g_resid, void* stream);
int nbdt_ref_bn_relu_pool(const float* x, const float* save_mean, const float* save_rstd, const float* gamma,
                          const float* beta, int32_t B, int32_t H, int32_t W, int32_t C, float* pooled, void* stream);
int nbdt_ref_stem_conv(const float* img, const float* w, int32_t B, int32_t H, int32_t W, int32_t cout_real,
                       int32_t cpad, int32_t stride, float* out, void* stream);
int nbdt_ref_stem_wgrad(const float* img, const float* gy, int32_t B, int32_t H, int32_t W, int32_t cout_real,
                        int32_t cpad, int32_t stride, float* dw, void* stream);
/* fp32-storage twins of the MBConv pieces (EfficientNet-B0; verification only, like everything in this block): same
 * arguments and meaning as nbdt_bn_act_apply / _pool / _bwd and nbdt_dwconv_fwd / _bwd_data / _bwd_weight.  The fused
 * forms of the product path have no twin: a depthwise forward leaves no statistics (nbdt_ref_bn_stats re-reads the
 * tensor), and the data gradient with BatchNorm-backward sums is nbdt_ref_dwconv_bwd_data followed by the full
 * nbdt_ref_bn_act_bwd.  Replaces, for verification, pytorchcv dwconv / BatchNorm2d + Swish / SEBlock scaling behind
 * nbdt/models/__init__.py:3. */
int nbdt_ref_bn_act_apply(const float* x, const float* save_mean, const float* save_rstd, const float* gamma,
                          const float* beta, int32_t act, const float* gate, const float* residual, int32_t B, int32_t H,
                          int32_t W, int32_t C, float* y, void* stream);
int nbdt_ref_bn_act_pool(const float* x, const float* save_mean, const float* save_rstd, const float* gamma,
                         const float* beta, int32_t act, const float* mul, float scale, int32_t B, int32_t H, int32_t W,
                         int32_t C, float* out, void* stream);
int nbdt_ref_bn_act_bwd(const float* gu, const float* gate, const float* gpool, const float* x, const float* save_mean,
                        const float* save_rstd, const float* gamma, const float* beta, int32_t act, const float* gx_add,
                        int32_t B, int32_t H, int32_t W, int32_t C, float* dsum, float* dgamma, float* dbeta, float* gx,
                        void* stream);
int nbdt_ref_dwconv_fwd(const float* x, const float* w, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                        int32_t stride, float* y, void* stream);
int nbdt_ref_dwconv_bwd_data(const float* gy, const float* w, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                             int32_t stride, float* gx, void* stream);
int nbdt_ref_dwconv_bwd_weight(const float* x, const float* gy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                               int32_t stride, float* dw, void* stream);

/* ------------------------------------------------------------------ stem / head / optimizer */
/* stem Conv2d(3->cout_real, 3x3, pad 1, stride 1|2) on NCHW fp32 images [B,3,H,W] -> padded NHWC
 * bf16 [B][H/stride+2][W/stride+2][cpad] (channels >= cout_real are zero).
 * w: fp32 [cout_real][3][3][3] (co, r, s, ci). */
int nbdt_stem_conv(const float* img, const float* w, int32_t B, int32_t H, int32_t W,
                   int32_t cout_real, int32_t cpad, int32_t stride, void* out, void* stream);
/* Weight gradient of that conv: dw[cout_real][3][3][3] (fp32) += sum over pixels of gy (padded NHWC bf16, channels
 * [0, cout_real)) x img.  cout_real must be a multiple of 8 and <= 72 (<= cpad): the kernel (round 4) reads the gradient
 * tile from LDS as float4 and keeps at most two 4-output groups per thread (every stem of the supported backbones is 16,
 * 32 or 64 wide).  Anything else returns NBDT_EINVAL -- rounds 1-3 accepted any cout_real <= 75. */
int nbdt_stem_wgrad(const float* img, const void* gy, int32_t B, int32_t H, int32_t W,
                    int32_t cout_real, int32_t cpad, int32_t stride, float* dw, void* stream);
/* nn.Linear: z[B][N] = x[B][K] w[N][K]^T + b (fp32) and its backward */
int nbdt_linear_fwd(const float* x, const float* w, const float* b, int32_t B, int32_t K, int32_t N,
                    float* z, void* stream);
int nbdt_linear_bwd(const float* x, const float* w, const float* gz, int32_t B, int32_t K, int32_t N,
                    float* gx, float* gw, float* gb, void* stream);
/* optim.SGD(momentum, weight_decay) over a flat fp32 buffer (main.py:207):
 * g = grad_scale*g + wd*p; buf = mom*buf + g; p -= lr*buf; optionally refreshes the bf16 copy the
 * conv kernels read (same element order as p) in the same pass; zero_grad != 0 also zeroes g (optimizer.zero_grad()
 * of the next step, main.py:235, folded into the same pass) */
int nbdt_sgd_step(float* p, float* g, float* buf, int64_t n, float lr, float momentum,
                  float weight_decay, float grad_scale, void* p_bf16, int32_t zero_grad, void* stream);

/* ------------------------------------------------------------------ measurement probe (not on the product path) */
/* A register-only stream of independent v_mfma_f32_32x32x16_bf16 on `blocks` CUs (one 512-thread block each, two waves
 * per SIMD): every wave issues iters x 16 of them (x 32768 flop).  bench.py times the launch for `roofline.mfma_stream`
 * -- the power / clock ceiling of the matrix pipes on the box the bench runs on (csrc/probe.hip).  sink: >= 1 float. */
int nbdt_probe_mfma_stream(int32_t blocks, int32_t iters, float* sink, void* stream);
/* LDS-operand MFMA streams: the dense K loop's ds_read_b128 + MFMA mix with no LDS-DMA, barriers or epilogue, for the
 * production tiling (variant 0: 8 waves x 64 pixels x 160 couts, 14 reads per 20 MFMAs) and for the one-wave-per-SIMD
 * tiling (variant 1: 4 waves x 128 pixels x 160 couts, 18 reads per 40 MFMAs, software-pipelined).  160 MFMAs per CU and
 * step either way: flops = blocks x iters x 160 x 32768.  Measurement only. */
int nbdt_probe_lds_mfma(int32_t blocks, int32_t iters, int32_t variant, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NBDT_HIP_H */
