"""Host-side launch helpers for the backbone kernels of libnbdt_hip.so.

Everything here is plumbing: tensor allocation (PyTorch caching allocator), tap-table construction
for the implicit-GEMM kernel and raw-pointer hand-off through ctypes.  No arithmetic on the data
path happens in Python/PyTorch.

Activation layout: padded NHWC bf16 ``[B][H+2][W+2][C]`` with a zero one-pixel border (kernels only
ever write interiors, so the border stays zero for the lifetime of the buffer).
Weight layout: ``[cout][taps][cin]`` (fp32 master and bf16 copy); the data-gradient kernel reads
the transposed, tap-reversed copy ``[cin][taps][cout]`` produced by ``weight_prep``.
"""
import ctypes

import torch

from nbdt import _C
from nbdt._C import ConvDesc, WgradDesc, check, lib, ptr

BN_SLOTS = 32
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def set_deterministic(on):
    """Process-wide: every cross-block reduction of the ResNet / WideResNet path in a fixed order instead of fp32
    atomics (nbdt_set_deterministic in include/nbdt_hip.h) -- same launches + same inputs => same bits, like the
    reference's CPU path.  Slower; for parity runs and debugging."""
    check(lib().nbdt_set_deterministic(1 if on else 0))


def is_deterministic():
    return bool(lib().nbdt_get_deterministic())


def set_wgrad_store_epilogue(on):
    """K-split weight gradient: 1 (default) = plain stores into per-split copies + a fold pass, 0 = fp32 atomics into dw
    (nbdt_set_wgrad_store_epilogue in include/nbdt_hip.h)."""
    check(lib().nbdt_set_wgrad_store_epilogue(1 if on else 0))


def wgrad_store_epilogue():
    return bool(lib().nbdt_get_wgrad_store_epilogue())


def set_reserved_cus(n):
    """CUs the one-block-per-CU MFMA kernels leave free for a collective's kernels (nbdt_set_reserved_cus)."""
    check(lib().nbdt_set_reserved_cus(int(n)))


def reserved_cus():
    return int(lib().nbdt_get_reserved_cus())


def padded(B, H, W, C, device, dtype=torch.bfloat16):
    """Zero-initialised padded NHWC activation buffer: bf16 (the product path), or fp32 for the engines'
    verification-only reference mode (see _ref below)."""
    return torch.zeros((B, H + 2, W + 2, C), dtype=dtype, device=device)


def _no_ref(t, what):
    if t.dtype == torch.float32:
        raise RuntimeError(f"fp32 reference mode has no twin of {what}")


def _ref(t):
    """fp32 padded tensors select the verification-only kernels of csrc/ref_fp32.hip (engine.set_reference_fp32): same
    operator, same descriptors, fp32 storage.  The product path never creates an fp32 activation buffer."""
    return t.dtype == torch.float32


def interior(t):
    """[B, H, W, C] view of the interior of a padded buffer."""
    return t[:, 1:-1, 1:-1, :]


_DEV_INDEX = {}


def stream_ptr(device):
    """Raw handle of torch's CURRENT stream on `device` (what every launch goes to).  torch._C._cuda_getCurrentRawStream is the
    one C call behind torch.cuda.current_stream(...).cuda_stream without the Stream object and the device-index resolution
    around it: a quarter of the host time of a ResNet18 / CIFAR10 step, which is bound by the launch path."""
    idx = _DEV_INDEX.get(device)
    if idx is None:
        d = torch.device(device)
        idx = _DEV_INDEX[device] = d.index if d.index is not None else torch.cuda.current_device()
    return ctypes.c_void_p(_raw_stream(idx))


def _raw_stream_public(idx):
    return torch.cuda.current_stream(idx).cuda_stream


# (a private binding: present in every torch this repo has met, but do not depend on it)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", _raw_stream_public)


def _fill(desc_arr, values):
    for i, v in enumerate(values):
        desc_arr[i] = int(v)


# ------------------------------------------------------------------------------------------------
# tap tables

# nbdt_conv_desc.ksplit of the descriptors built below (0 automatic, 1 never split K): A/B measurements set it before
# an engine plans its launches
CONV_KSPLIT = {"fwd": 0, "dgrad": 0}


def conv_fwd_desc(B, Hi, Wi, cin, cout, k, stride):
    """Conv2d(k in {1,3}, padding=k//2, stride) forward: in [B,Hi,Wi,cin] -> out [B,Hi/s,Wi/s,cout]."""
    assert k in (1, 3) and Hi % stride == 0 and Wi % stride == 0
    Ho, Wo = Hi // stride, Wi // stride
    d = ConvDesc()
    d.B, d.gh, d.gw, d.cin, d.cout = B, Ho, Wo, cin, cout
    rowi = (Wi + 2) * cin
    if k == 3:
        d.ntaps = d.w_ntaps = 9
        _fill(d.tap_off, [r * rowi + s * cin for r in range(3) for s in range(3)])
        _fill(d.w_tap, range(9))
        d.in_base = 0
    else:
        d.ntaps = d.w_ntaps = 1
        d.tap_off[0] = 0
        d.w_tap[0] = 0
        d.in_base = rowi + cin
    d.in_bs, d.in_hs, d.in_ws = (Hi + 2) * rowi, stride * rowi, stride * cin
    rowo = (Wo + 2) * cout
    d.out_bs, d.out_hs, d.out_ws, d.out_base = (Ho + 2) * rowo, rowo, cout, rowo + cout
    d.accumulate = 0
    d.wide_tile = 1      # forward launches run alone on the GPU (data gradients share it with weight gradients)
    d.ksplit = CONV_KSPLIT["fwd"]
    return d


def conv_dgrad_descs(B, Hi, Wi, cin, cout, k, stride, accumulate=False):
    """Data gradient of the conv above: g_out [B,Ho,Wo,cout] -> g_in [B,Hi,Wi,cin].

    Returns a list of launches (1 for stride 1; 4 output-parity classes for a strided 3x3).  The
    weight operand is the tap-reversed transposed copy ``wd[cin][taps][cout]``.
    """
    assert k in (1, 3) and stride in (1, 2)
    Ho, Wo = Hi // stride, Wi // stride
    rowg = (Wo + 2) * cout          # gradient (kernel input) row, channels = cout
    rowx = (Wi + 2) * cin           # g_in (kernel output) row, channels = cin
    out = []
    if k == 1:
        d = ConvDesc()
        d.B, d.gh, d.gw, d.cin, d.cout = B, Ho, Wo, cout, cin
        d.ntaps = d.w_ntaps = 1
        d.tap_off[0] = 0
        d.w_tap[0] = 0
        d.in_bs, d.in_hs, d.in_ws, d.in_base = (Ho + 2) * rowg, rowg, cout, rowg + cout
        d.out_bs, d.out_hs, d.out_ws, d.out_base = (Hi + 2) * rowx, stride * rowx, stride * cin, rowx + cin
        d.accumulate = 1 if accumulate else 0
        if stride == 2 and not accumulate:
            raise ValueError("a strided 1x1 dgrad only touches every other pixel: use accumulate=True "
                             "into a fully written buffer")
        return [d]
    if stride == 1:
        d = ConvDesc()
        d.B, d.gh, d.gw, d.cin, d.cout = B, Hi, Wi, cout, cin
        d.ntaps = d.w_ntaps = 9
        _fill(d.tap_off, [r * rowg + s * cout for r in range(3) for s in range(3)])
        _fill(d.w_tap, range(9))     # wd is already tap-reversed
        d.in_bs, d.in_hs, d.in_ws, d.in_base = (Ho + 2) * rowg, rowg, cout, 0
        d.out_bs, d.out_hs, d.out_ws, d.out_base = (Hi + 2) * rowx, rowx, cin, rowx + cin
        d.accumulate = 1 if accumulate else 0
        d.ksplit = CONV_KSPLIT["dgrad"]
        return [d]
    # stride 2, 3x3: padded input row hp = 2*ho + r.  hp odd (=2i+1): r=1, ho=i.
    # hp even (=2i+2): r=0 -> ho=i+1 ; r=2 -> ho=i.  Same for columns.  g rows are padded (+1).
    taps_1d = {1: [(1, 1)], 0: [(0, 2), (2, 1)]}   # parity -> [(r, padded g row shift)]
    for ph in (1, 0):
        for pw in (1, 0):
            d = ConvDesc()
            d.B, d.gh, d.gw, d.cin, d.cout = B, Ho, Wo, cout, cin
            taps = [(r, sr, s, sc) for (r, sr) in taps_1d[ph] for (s, sc) in taps_1d[pw]]
            d.ntaps, d.w_ntaps = len(taps), 9
            _fill(d.tap_off, [sr * rowg + sc * cout for (_, sr, _, sc) in taps])
            _fill(d.w_tap, [8 - (3 * r + s) for (r, _, s, _) in taps])
            d.in_bs, d.in_hs, d.in_ws, d.in_base = (Ho + 2) * rowg, rowg, cout, 0
            hp0, wp0 = (1 if ph else 2), (1 if pw else 2)
            d.out_bs, d.out_hs, d.out_ws = (Hi + 2) * rowx, 2 * rowx, 2 * cin
            d.out_base = hp0 * rowx + wp0 * cin
            d.accumulate = 1 if accumulate else 0
            out.append(d)
    return out


CONV_FORMS = {0: "conv_igemm_dma", 1: "4-wave 256-pixel", 2: "ping-pong 512-pixel", 3: "ping-pong 512-pixel, padded pitch",
              4: "ping-pong 256-pixel half tiles"}


def conv_plan(desc):
    """(form, ksplit) a launch of this descriptor would take -- host only (nbdt_conv_plan in include/nbdt_hip.h)."""
    form, ks = ctypes.c_int32(0), ctypes.c_int32(0)
    check(lib().nbdt_conv_plan(ctypes.byref(desc), ctypes.byref(form), ctypes.byref(ks)))
    return form.value, ks.value


def conv_wgrad_desc(B, Hi, Wi, cin, cout, k, stride):
    Ho, Wo = Hi // stride, Wi // stride
    d = WgradDesc()
    d.B, d.gh, d.gw, d.cin, d.cout = B, Ho, Wo, cin, cout
    rowi = (Wi + 2) * cin
    if k == 3:
        d.ntaps = d.w_ntaps = 9
        _fill(d.tap_off, [r * rowi + s * cin for r in range(3) for s in range(3)])
        _fill(d.w_tap, range(9))
        d.x_base = 0
    else:
        d.ntaps = d.w_ntaps = 1
        d.tap_off[0] = 0
        d.w_tap[0] = 0
        d.x_base = rowi + cin
    d.x_bs, d.x_hs, d.x_ws = (Hi + 2) * rowi, stride * rowi, stride * cin
    rowo = (Wo + 2) * cout
    d.g_bs, d.g_hs, d.g_ws, d.g_base = (Ho + 2) * rowo, rowo, cout, rowo + cout
    return d


# ------------------------------------------------------------------------------------------------
# optional per-launch timing of the MFMA kernels with HIP events on the launch stream (bench.py)

class KernelTimer:
    """Brackets every conv_igemm / conv_wgrad launch with HIP events on the stream the kernel is
    launched on (torch's current stream) and accumulates algorithmic flops per kernel family."""

    def __init__(self, only=None):
        self.items = []   # (kind, flops, start_event, end_event)
        self.only = only  # restrict to these kinds (None: all)
        self.kernels = {}  # device kernel name (nbdt_debug_last_igemm / _wgrad) -> launches bracketed

    def wants(self, kind):
        return self.only is None or kind in self.only

    def bracket(self, kind, flops, device):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        self.items.append((kind, flops, s, e))
        return s, e

    def count(self, kernel_name):
        self.kernels[kernel_name] = self.kernels.get(kernel_name, 0) + 1

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, flops, s, e in self.items:
            d = out.setdefault(kind, {"launches": 0, "flops": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += s.elapsed_time(e)
        for d in out.values():
            d["avg_us"] = 1e3 * d["ms"] / max(d["launches"], 1)
            d["tflops"] = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        return out


_timer = None


def set_timer(t):
    global _timer
    _timer = t


# ------------------------------------------------------------------------------------------------
# launches

def conv_igemm(desc, inp, w_bf16, out, residual=None, bn_scratch=None):
    """bn_scratch: also accumulate the output's per-channel sum / sum-of-squares for the next BatchNorm."""
    if _ref(inp):
        check(lib().nbdt_ref_conv(ctypes.byref(desc), ptr(inp), ptr(w_bf16), ptr(out), ptr(residual),
                                  stream_ptr(inp.device)))
        if bn_scratch is not None:      # the epilogue's statistics: row 0 of the partial table, folded by bn_finalize
            B, H, W, C = _dims(out)
            check(lib().nbdt_ref_bn_stats(ptr(out), B, H, W, C, BN_EPS, BN_MOMENTUM, None, None, None, None,
                                          ptr(bn_scratch), stream_ptr(inp.device)))
        return
    ev = None
    if _timer is not None and _timer.wants("conv_igemm"):
        flops = desc_flops(desc)
        ev = _timer.bracket("conv_igemm", flops, inp.device)
        ev[0].record()
    if bn_scratch is None:
        check(lib().nbdt_conv_igemm(ctypes.byref(desc), ptr(inp), ptr(w_bf16), ptr(out), ptr(residual),
                                    stream_ptr(inp.device)))
    else:
        check(lib().nbdt_conv_igemm_stats(ctypes.byref(desc), ptr(inp), ptr(w_bf16), ptr(out), ptr(residual),
                                          ptr(bn_scratch), stream_ptr(inp.device)))
    if ev is not None:
        ev[1].record()
        _timer.count(last_igemm_kernel_full())


def desc_flops(desc):
    """Algorithmic flops of one conv / data-gradient / weight-gradient launch: 2 x pixels x taps x cin x cout with the
    layer's REAL channel counts when the engine attached them (desc.flop_channels; the kernels multiply channels padded
    to 32 -- WRN's first unit 16 -> 32 -- and round 4's roofline counted those, +0.5 %), else the descriptor's own."""
    cin, cout = getattr(desc, "flop_channels", (desc.cin, desc.cout))
    return 2.0 * desc.B * desc.gh * desc.gw * desc.ntaps * cin * cout


def conv_igemm_multi(descs, inp, w_bf16, out):
    """The launches of `descs` (<= 4, same operands: the parity classes of a strided 3x3 data gradient) in one grid."""
    if _ref(inp):
        for d in descs:
            check(lib().nbdt_ref_conv(ctypes.byref(d), ptr(inp), ptr(w_bf16), ptr(out), None, stream_ptr(inp.device)))
        return
    arr = (ConvDesc * len(descs))(*descs)
    ev = None
    if _timer is not None and _timer.wants("conv_igemm"):
        flops = sum(desc_flops(d) for d in descs)
        ev = _timer.bracket("conv_igemm", flops, inp.device)
        ev[0].record()
    check(lib().nbdt_conv_igemm_multi(arr, len(descs), ptr(inp), ptr(w_bf16), ptr(out), stream_ptr(inp.device)))
    if ev is not None:
        ev[1].record()
        _timer.count(last_igemm_kernel_full())


def conv_igemm_bnbwd(desc, inp, w_bf16, out, bn_x, mean, rstd, gamma, beta, partials):
    """dgrad launch that also produces the BatchNorm-backward sums of (out, bn_x) as per-tile partials."""
    _no_ref(inp, "conv_igemm_bnbwd (fused BatchNorm-backward sums): use the split or the unfused schedule")
    ev = None
    if _timer is not None and _timer.wants("conv_igemm"):
        flops = desc_flops(desc)
        ev = _timer.bracket("conv_igemm", flops, inp.device)
        ev[0].record()
    check(lib().nbdt_conv_igemm_bnbwd(ctypes.byref(desc), ptr(inp), ptr(w_bf16), ptr(out), ptr(bn_x), ptr(mean),
                                      ptr(rstd), ptr(gamma), ptr(beta), ptr(partials), stream_ptr(inp.device)))
    if ev is not None:
        ev[1].record()
        _timer.count(last_igemm_kernel_full())


def conv_igemm_affine(desc, inp, w_bf16, out, scale, shift, act=1, residual=None):
    """Inference launch: out = act(conv * scale[c] + shift[c] [+ residual]) (act: 0 none, 1 ReLU, 2 swish)."""
    _no_ref(inp, "conv_igemm_affine (folded eval-mode BatchNorm): engine.fuse_eval is off in reference mode")
    check(lib().nbdt_conv_igemm_affine(ctypes.byref(desc), ptr(inp), ptr(w_bf16), ptr(out), ptr(residual),
                                       ptr(scale), ptr(shift), act, stream_ptr(inp.device)))


def bn_bwd_fused(gy, x, mean, rstd, gamma, beta, partials, dsum, dgamma, dbeta, gx, gx_add=None, cus=0):
    """BatchNorm(+ReLU) backward when the producing dgrad already left the reduction partials.
    cus > 0: the apply pass runs on that many CUs only (nbdt_bn_bwd_apply_cus) -- the caller has a weight gradient
    with cu_budget = 256 - cus in flight on another stream."""
    _no_ref(x, "bn_bwd_fused: use the split or the unfused schedule")
    B, H, W, C = _dims(x)
    st = stream_ptr(x.device)
    check(lib().nbdt_bn_bwd_fold(B, H, W, C, ptr(partials), ptr(dsum), ptr(dgamma), ptr(dbeta), st))
    if cus > 0:
        check(lib().nbdt_bn_bwd_apply_cus(ptr(gy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dsum),
                                          ptr(gx_add), B, H, W, C, ptr(gx), int(cus), st))
        return
    check(lib().nbdt_bn_bwd_apply(ptr(gy), None, ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                                  ptr(dsum), ptr(gx_add), 1, B, H, W, C, ptr(gx), None, st))


def bn_bwd_cus(gy, x, mean, rstd, gamma, beta, scratch, dsum, dgamma, dbeta, gx, cus, gx_add=None):
    """Whole BatchNorm(+ReLU) backward -- sums, fold, elementwise pass -- on `cus` CUs (no partials from a dgrad
    epilogue needed).  scratch: one zeroed 32-slot buffer (nbdt_bn_bwd_reduce_cus + nbdt_bn_bwd_apply_cus, three
    launches), or a PAIR (slots, slots_other) of them: nbdt_bn_bwd_cus, two launches -- the fold runs in the prologue of
    the elementwise pass; `slots` is left dirty and `slots_other` zeroed, so the caller swaps them for the next call."""
    B, H, W, C = _dims(x)
    st = stream_ptr(x.device)
    if _ref(x):
        check(lib().nbdt_ref_bn_bwd(ptr(gy), None, None, ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), 1,
                                    ptr(gx_add), B, H, W, C, 1, ptr(dsum), ptr(dgamma), ptr(dbeta), ptr(gx), None, st))
        return
    if isinstance(scratch, (tuple, list)):
        slots, other = scratch
        check(lib().nbdt_bn_bwd_cus(ptr(gy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(gx_add), B, H, W,
                                    C, ptr(slots), ptr(other), ptr(dsum), ptr(dgamma), ptr(dbeta), ptr(gx), int(cus),
                                    st))
        return
    check(lib().nbdt_bn_bwd_reduce_cus(ptr(gy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), B, H, W, C,
                                       ptr(scratch), ptr(dsum), ptr(dgamma), ptr(dbeta), int(cus), st))
    check(lib().nbdt_bn_bwd_apply_cus(ptr(gy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dsum),
                                      ptr(gx_add), B, H, W, C, ptr(gx), int(cus), st))


def conv_wgrad(desc, x, gy, dw, cu_budget=0):
    """cu_budget: size the launch for that many CUs (0 = all) -- see nbdt_wgrad_desc.cu_budget."""
    desc.cu_budget = int(cu_budget)
    if _ref(x):
        check(lib().nbdt_ref_wgrad(ctypes.byref(desc), ptr(x), ptr(gy), ptr(dw), stream_ptr(x.device)))
        return
    ev = None
    if _timer is not None and _timer.wants("conv_wgrad"):
        flops = desc_flops(desc)
        ev = _timer.bracket("conv_wgrad", flops, x.device)
        ev[0].record()
    check(lib().nbdt_conv_wgrad(ctypes.byref(desc), ptr(x), ptr(gy), ptr(dw), stream_ptr(x.device)))
    if ev is not None:
        ev[1].record()


def conv_wgrad_blocks(desc, cu_budget=0):
    """Thread blocks (= CUs, for the 8-wave kernel) the weight-gradient launch of `desc` will use with that budget;
    0 when the launch is not one of the CU-budgeted dense 3x3 kernels."""
    desc.cu_budget = int(cu_budget)
    return int(lib().nbdt_conv_wgrad_blocks(ctypes.byref(desc)))


MI355X_CUS, MI355X_XCDS = 256, 8


def cu_topology_is_mi355x(device):
    """The CU-sharing arithmetic below is MI355X's: 256 CUs in 8 XCDs of 32, thread blocks dealt to the XCDs
    round-robin.  Anything else (MI300X: 304 CUs, 38 per XCD) gets no sharing rather than a wrong split."""
    props = torch.cuda.get_device_properties(device)
    return props.multi_processor_count == MI355X_CUS and "gfx950" in getattr(props, "gcnArchName", "gfx950")


def plan_cu_share(desc, elements, tensors, gbps_per_cu, target_us, min_cus, max_cus):
    """(CU budget of the weight gradient of `desc`, CUs of the HBM-bound pass of `tensors` tensors of `elements` bf16
    that runs beside it).  n = bytes / (gbps_per_cu x target_us), clamped to [max(8, min_cus), max_cus]; then the XCD
    rule: thread blocks go to the 8 XCDs round-robin by block index, per kernel -- the weight gradient's blocks (a
    whole number of pixel splits per (cout, cin) tile: 205 for a budget of 208, 160 for 232 with 80 tiles) put
    ceil(blocks / 8) on each XCD, so the pass may take 32 minus that on EACH.  One block more on any XCD waits for a
    weight-gradient block to finish there and the pass takes as long as both (measured: 19.1 -> 27 ms per step
    whenever the dispatch order fell that way).  A budget whose blocks would leave the pass less than one CU per XCD
    is lowered by 8 until they do, so the pass never gets 0 CUs.  Host arithmetic only."""
    held = (reserved_cus() + MI355X_XCDS - 1) // MI355X_XCDS      # per XCD: CUs a collective's kernels hold right now
    per_xcd = MI355X_CUS // MI355X_XCDS - held
    cus = per_xcd * MI355X_XCDS                                    # what the pair has to share
    n = int(round(elements * 2 * tensors / (gbps_per_cu * 1e9 * target_us * 1e-6)))
    n = max(max(8, int(min_cus)), min(int(max_cus), n))
    n = min(n, cus - MI355X_XCDS)
    while True:
        blocks = conv_wgrad_blocks(desc, cus - n)
        if not 0 < blocks <= cus - n:             # not the one-block-per-CU kernel (small problems): the model's n
            return cus - n, n
        free = MI355X_XCDS * (per_xcd - (blocks + MI355X_XCDS - 1) // MI355X_XCDS)
        if free >= MI355X_XCDS or n + MI355X_XCDS > cus - MI355X_XCDS:
            return cus - n, max(free, MI355X_XCDS)
        n += MI355X_XCDS


def weight_prep(w_fp32, cout, taps, cin, w_bf16=None, wd_bf16=None):
    check(lib().nbdt_weight_prep(ptr(w_fp32), cout, taps, cin, ptr(w_bf16), ptr(wd_bf16),
                                 stream_ptr(w_fp32.device)))


def weight_tile_batched(src_bf16, table, n, total, dst_bf16):
    check(lib().nbdt_weight_tile_batched(ptr(src_bf16), ptr(table), n, total, ptr(dst_bf16),
                                         stream_ptr(src_bf16.device)))


def weight_tiles(w_bf16):
    """One [rows][9][k] bf16 weight matrix (forward weights, or the transposed data-gradient copy) re-stored as the
    DMA-ordered tiles the dense 3x3 kernels read (desc.w_tiled): the single-matrix form of weight_tile_batched."""
    rows, taps, k = w_bf16.shape
    assert taps == 9 and rows % 32 == 0 and k % 32 == 0
    r32 = rows // 32
    nt = 5 if r32 % 5 == 0 else 4 if r32 % 4 == 0 else 2 if r32 % 2 == 0 else 1
    total = (rows // (32 * nt)) * (k // 32) * 9
    table = torch.tensor([[0, 0, rows, k, 0]], dtype=torch.int64, device=w_bf16.device)
    dst = torch.empty(rows * 9 * k, dtype=torch.bfloat16, device=w_bf16.device)
    weight_tile_batched(w_bf16.contiguous(), table, 1, total, dst)
    return dst


def last_igemm_kernel():
    """Name of the device kernel this thread's last conv_igemm* call launched (tests assert on it)."""
    return lib().nbdt_debug_last_igemm().decode()


def last_igemm_kernel_full():
    """... with its template arguments, as rocprofv3 prints the device kernel (bench.py's traffic guard)."""
    return lib().nbdt_debug_last_igemm_full().decode()


def last_wgrad_kernel():
    return lib().nbdt_debug_last_wgrad().decode()


def weight_prep_batched(flat, table, n_layers, total, wd_flat):
    check(lib().nbdt_weight_prep_batched(ptr(flat), ptr(table), n_layers, total, ptr(wd_flat),
                                         stream_ptr(flat.device)))


def _dims(t):
    B, Hp, Wp, C = t.shape
    return B, Hp - 2, Wp - 2, C


def bn_stats(x, scratch, save_mean, save_rstd, running_mean=None, running_var=None,
             eps=BN_EPS, momentum=BN_MOMENTUM, slots_filled=False):
    """slots_filled: the producing kernel already accumulated the sums into `scratch` (fold only)."""
    B, H, W, C = _dims(x)
    if _ref(x):
        check(lib().nbdt_ref_bn_stats(ptr(x), B, H, W, C, eps, momentum, ptr(running_mean), ptr(running_var),
                                      ptr(save_mean), ptr(save_rstd), None, stream_ptr(x.device)))
        return
    check(lib().nbdt_bn_stats(None if slots_filled else ptr(x), B, H, W, C, eps, momentum, ptr(running_mean), ptr(running_var),
                              ptr(scratch), ptr(save_mean), ptr(save_rstd), stream_ptr(x.device)))


def bn_finalize(x, scratch, save_mean, save_rstd, running_mean=None, running_var=None,
                eps=BN_EPS, momentum=BN_MOMENTUM):
    """Fold sums that a conv epilogue already left in `scratch` (conv_igemm(..., bn_scratch=scratch))."""
    B, H, W, C = _dims(x)
    check(lib().nbdt_bn_finalize(B, H, W, C, eps, momentum, ptr(running_mean), ptr(running_var), ptr(scratch),
                                 ptr(save_mean), ptr(save_rstd), stream_ptr(x.device)))


def bn_apply(x, mean, rstd, gamma, beta, y, relu=True, residual=None):
    B, H, W, C = _dims(x)
    if _ref(x):
        check(lib().nbdt_ref_bn_apply(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(residual),
                                      1 if relu else 0, B, H, W, C, ptr(y), stream_ptr(x.device)))
        return
    check(lib().nbdt_bn_apply(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(residual),
                              1 if relu else 0, B, H, W, C, ptr(y), stream_ptr(x.device)))


def bn_bwd(gy, y, x, mean, rstd, gamma, scratch, dsum, dgamma, dbeta, gx, relu=True, gx_add=None,
           g_resid=None, beta=None):
    """y=None (allowed when the forward had no residual): the ReLU mask is recomputed from x/gamma/beta."""
    B, H, W, C = _dims(x)
    st = stream_ptr(x.device)
    r = 1 if relu else 0
    if _ref(x):
        check(lib().nbdt_ref_bn_bwd(ptr(gy), None, ptr(y), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), r,
                                    ptr(gx_add), B, H, W, C, 1, ptr(dsum), ptr(dgamma), ptr(dbeta), ptr(gx), ptr(g_resid),
                                    st))
        return
    check(lib().nbdt_bn_bwd_reduce(ptr(gy), ptr(y), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), r,
                                   B, H, W, C, ptr(scratch), ptr(dsum), ptr(dgamma), ptr(dbeta), st))
    check(lib().nbdt_bn_bwd_apply(ptr(gy), ptr(y), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                                  ptr(dsum), ptr(gx_add), r, B, H, W, C, ptr(gx), ptr(g_resid), st))


def bn_relu_pool(x, mean, rstd, gamma, beta, pooled):
    B, H, W, C = _dims(x)
    if _ref(x):
        check(lib().nbdt_ref_bn_relu_pool(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), B, H, W, C, ptr(pooled),
                                          stream_ptr(x.device)))
        return
    check(lib().nbdt_bn_relu_pool(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), B, H, W, C,
                                  ptr(pooled), stream_ptr(x.device)))


def pool_bn_bwd(gpooled, x, mean, rstd, gamma, beta, scratch, dsum, dgamma, dbeta, gx):
    B, H, W, C = _dims(x)
    st = stream_ptr(x.device)
    if _ref(x):
        check(lib().nbdt_ref_bn_bwd(None, ptr(gpooled), None, ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), 1,
                                    None, B, H, W, C, 1, ptr(dsum), ptr(dgamma), ptr(dbeta), ptr(gx), None, st))
        return
    check(lib().nbdt_pool_bn_bwd_reduce(ptr(gpooled), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                                        B, H, W, C, ptr(scratch), ptr(dsum), ptr(dgamma), ptr(dbeta), st))
    check(lib().nbdt_pool_bn_bwd_apply(ptr(gpooled), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                                       ptr(dsum), B, H, W, C, ptr(gx), st))


def pool_bn_bwd_apply(gpooled, x, mean, rstd, gamma, beta, dsum, gx):
    """Elementwise pass of the pooled head's backward alone, with the caller's sums (ResNetEngine: a plain average pool
    is an identity BatchNorm with zero batch sums)."""
    B, H, W, C = _dims(x)
    st = stream_ptr(x.device)
    if _ref(x):
        check(lib().nbdt_ref_bn_bwd(None, ptr(gpooled), None, ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), 1,
                                    None, B, H, W, C, 0, ptr(dsum), None, None, ptr(gx), None, st))
        return
    check(lib().nbdt_pool_bn_bwd_apply(ptr(gpooled), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dsum),
                                       B, H, W, C, ptr(gx), st))


def stem_conv(img, w, out, cout_real, stride=1):
    B, _, H, W = img.shape
    if _ref(out):
        check(lib().nbdt_ref_stem_conv(ptr(img), ptr(w), B, H, W, cout_real, out.shape[3], stride, ptr(out),
                                       stream_ptr(img.device)))
        return
    check(lib().nbdt_stem_conv(ptr(img), ptr(w), B, H, W, cout_real, out.shape[3], stride, ptr(out),
                               stream_ptr(img.device)))


def stem_wgrad(img, gy, dw, cout_real, stride=1):
    B, _, H, W = img.shape
    if _ref(gy):
        check(lib().nbdt_ref_stem_wgrad(ptr(img), ptr(gy), B, H, W, cout_real, gy.shape[3], stride, ptr(dw),
                                        stream_ptr(img.device)))
        return
    check(lib().nbdt_stem_wgrad(ptr(img), ptr(gy), B, H, W, cout_real, gy.shape[3], stride, ptr(dw),
                                stream_ptr(img.device)))


def linear_fwd(x, w, b, z):
    B, K = x.shape
    N = w.shape[0]
    check(lib().nbdt_linear_fwd(ptr(x), ptr(w), ptr(b), B, K, N, ptr(z), stream_ptr(x.device)))


def linear_bwd(x, w, gz, gx, gw, gb):
    B, K = x.shape
    N = w.shape[0]
    check(lib().nbdt_linear_bwd(ptr(x), ptr(w), ptr(gz), B, K, N, ptr(gx), ptr(gw), ptr(gb),
                                stream_ptr(x.device)))


def sgd_step(p, g, buf, lr, momentum, weight_decay, grad_scale=1.0, p_bf16=None, zero_grad=False):
    """zero_grad: the same pass leaves g zeroed (the next step's zero_grad())."""
    check(lib().nbdt_sgd_step(ptr(p), ptr(g), ptr(buf), p.numel(), lr, momentum, weight_decay, grad_scale,
                              ptr(p_bf16), 1 if zero_grad else 0, stream_ptr(p.device)))


# ------------------------------------------------------------------------------------------------
# MBConv pieces (EfficientNet-B0, csrc/effnet.hip)

ACT_NONE, ACT_RELU, ACT_SWISH = 0, 1, 2


def bn_act_apply(x, mean, rstd, gamma, beta, y, act=ACT_SWISH, gate=None, residual=None):
    B, H, W, C = _dims(x)
    if _ref(x):
        check(lib().nbdt_ref_bn_act_apply(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, ptr(gate),
                                          ptr(residual), B, H, W, C, ptr(y), stream_ptr(x.device)))
        return
    check(lib().nbdt_bn_act_apply(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, ptr(gate),
                                  ptr(residual), B, H, W, C, ptr(y), stream_ptr(x.device)))


def bn_act_pool(x, mean, rstd, gamma, beta, out, act=ACT_SWISH, mul=None, scale=None):
    B, H, W, C = _dims(x)
    scale = 1.0 / (H * W) if scale is None else scale
    if _ref(x):
        check(lib().nbdt_ref_bn_act_pool(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, ptr(mul), scale,
                                         B, H, W, C, ptr(out), stream_ptr(x.device)))
        return
    check(lib().nbdt_bn_act_pool(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, ptr(mul), scale,
                                 B, H, W, C, ptr(out), stream_ptr(x.device)))


def bn_act_bwd(gu, x, mean, rstd, gamma, beta, scratch, dsum, dgamma, dbeta, gx, act=ACT_SWISH, gate=None,
               gpool=None, gx_add=None):
    B, H, W, C = _dims(x)
    if _ref(x):
        check(lib().nbdt_ref_bn_act_bwd(ptr(gu), ptr(gate), ptr(gpool), ptr(x), ptr(mean), ptr(rstd), ptr(gamma),
                                        ptr(beta), act, ptr(gx_add), B, H, W, C, ptr(dsum), ptr(dgamma), ptr(dbeta),
                                        ptr(gx), stream_ptr(x.device)))
        return
    check(lib().nbdt_bn_act_bwd(ptr(gu), ptr(gate), ptr(gpool), ptr(x), ptr(mean), ptr(rstd), ptr(gamma),
                                ptr(beta), act, ptr(gx_add), B, H, W, C, ptr(scratch), ptr(dsum), ptr(dgamma),
                                ptr(dbeta), ptr(gx), stream_ptr(x.device)))


def dwconv_fwd(x, w, y, k, stride, bn_scratch=None):
    """bn_scratch: the 32-slot BN scratch; the kernel adds sum(y), sum(y^2) (fold with bn_stats(None, ...))."""
    B, H, W, C = _dims(x)
    if _ref(x):    # (no statistics from the fp32 twin: bn_stats re-reads the tensor in reference mode)
        check(lib().nbdt_ref_dwconv_fwd(ptr(x), ptr(w), B, H, W, C, k, stride, ptr(y), stream_ptr(x.device)))
        return
    check(lib().nbdt_dwconv_fwd(ptr(x), ptr(w), B, H, W, C, k, stride, ptr(y), ptr(bn_scratch),
                                stream_ptr(x.device)))


def dwconv_bwd_data(gy, w, gx, k, stride):
    B, H, W, C = _dims(gx)
    if _ref(gx):
        check(lib().nbdt_ref_dwconv_bwd_data(ptr(gy), ptr(w), B, H, W, C, k, stride, ptr(gx), stream_ptr(gx.device)))
        return
    check(lib().nbdt_dwconv_bwd_data(ptr(gy), ptr(w), B, H, W, C, k, stride, ptr(gx), stream_ptr(gx.device)))


def dwconv_bwd_data_bn(gy, w, gx, k, bn_x, mean, rstd, gamma, beta, scratch):
    """Stride-1 depthwise data gradient + the backward sums of the BatchNorm + swish that produced its input (into the
    32-slot scratch); follow with bn_act_bwd_apply."""
    B, H, W, C = _dims(gx)
    if _ref(gx):   # the plain data gradient; the sums are recomputed by bn_act_bwd_apply's reference form
        check(lib().nbdt_ref_dwconv_bwd_data(ptr(gy), ptr(w), B, H, W, C, k, 1, ptr(gx), stream_ptr(gx.device)))
        return
    check(lib().nbdt_dwconv_bwd_data_bn(ptr(gy), ptr(w), B, H, W, C, k, ptr(gx), ptr(bn_x), ptr(mean), ptr(rstd),
                                        ptr(gamma), ptr(beta), ptr(scratch), stream_ptr(gx.device)))


def bn_act_bwd_apply(gu, x, mean, rstd, gamma, beta, scratch, dsum, dgamma, dbeta, gx, act=ACT_SWISH, gx_add=None):
    """bn_act_bwd without its reduction pass (the producing kernel filled the slots)."""
    B, H, W, C = _dims(x)
    if _ref(x):    # reference mode: sums + elementwise pass (dwconv_bwd_data_bn's twin left no sums)
        check(lib().nbdt_ref_bn_act_bwd(ptr(gu), None, None, ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act,
                                        ptr(gx_add), B, H, W, C, ptr(dsum), ptr(dgamma), ptr(dbeta), ptr(gx),
                                        stream_ptr(x.device)))
        return
    check(lib().nbdt_bn_act_bwd_apply(ptr(gu), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, ptr(gx_add),
                                      B, H, W, C, ptr(scratch), ptr(dsum), ptr(dgamma), ptr(dbeta), ptr(gx),
                                      stream_ptr(x.device)))


def bn_act_se_sums(gu, x, mean, rstd, gamma, beta, sums, act=ACT_SWISH):
    """One pass over (gu, x): sums[5][B][C] = dL/dgate and the four per-(image, channel) sums the BatchNorm backward of the
    SE-scaled activation is linear in (include/nbdt_hip.h: nbdt_bn_act_se_sums).  bf16 production path only."""
    B, H, W, C = _dims(x)
    check(lib().nbdt_bn_act_se_sums(ptr(gu), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, B, H, W, C,
                                    ptr(sums), stream_ptr(x.device)))


def bn_act_se_bwd_apply(gu, gate, gpool, sums, x, mean, rstd, gamma, beta, dsum, dgamma, dbeta, gx, act=ACT_SWISH):
    """Fold bn_act_se_sums' sums with gate / gpool into the BatchNorm-backward sums, then bn_act_bwd's elementwise pass."""
    B, H, W, C = _dims(x)
    check(lib().nbdt_bn_act_se_bwd_apply(ptr(gu), ptr(gate), ptr(gpool), ptr(sums), ptr(x), ptr(mean), ptr(rstd),
                                         ptr(gamma), ptr(beta), act, B, H, W, C, ptr(dsum), ptr(dgamma), ptr(dbeta),
                                         ptr(gx), stream_ptr(x.device)))


def dwconv_bwd_weight(x, gy, dw, k, stride):
    B, H, W, C = _dims(x)
    if _ref(x):
        check(lib().nbdt_ref_dwconv_bwd_weight(ptr(x), ptr(gy), B, H, W, C, k, stride, ptr(dw), stream_ptr(x.device)))
        return
    check(lib().nbdt_dwconv_bwd_weight(ptr(x), ptr(gy), B, H, W, C, k, stride, ptr(dw), stream_ptr(x.device)))


def se_gate_fwd(pooled, w1, b1, w2, b2, pre1, gate, c_real):
    B, C = pooled.shape
    check(lib().nbdt_se_gate_fwd(ptr(pooled), ptr(w1), ptr(b1), ptr(w2), ptr(b2), B, C, c_real, w1.shape[0],
                                 ptr(pre1), ptr(gate), stream_ptr(pooled.device)))


def se_gate_bwd(dgate, gate, pre1, pooled, w1, w2, dpre2, dpre1, gpool, dw1, db1, dw2, db2, c_real):
    """dw1 = db1 = dw2 = db2 = None: the data part only (follow with se_param_grad, on any stream ordered after this)."""
    B, C = pooled.shape
    check(lib().nbdt_se_gate_bwd(ptr(dgate), ptr(gate), ptr(pre1), ptr(pooled), ptr(w1), ptr(w2), B, C, c_real,
                                 w1.shape[0], ptr(dpre2), ptr(dpre1), ptr(gpool), ptr(dw1), ptr(db1), ptr(dw2),
                                 ptr(db2), stream_ptr(pooled.device)))


def se_param_grad(dpre2, dpre1, pre1, pooled, dw1, db1, dw2, db2, c_real):
    B, C = pooled.shape
    check(lib().nbdt_se_param_grad(ptr(dpre2), ptr(dpre1), ptr(pre1), ptr(pooled), B, C, c_real, dw1.shape[0],
                                   ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), stream_ptr(pooled.device)))


def dropout_fwd(x, p, seed, mask, y):
    check(lib().nbdt_dropout_fwd(ptr(x), x.numel(), p, seed & 0xffffffff, ptr(mask), ptr(y), stream_ptr(x.device)))


def dropout_bwd(gy, p, mask, gx):
    check(lib().nbdt_dropout_bwd(ptr(gy), gy.numel(), p, ptr(mask), ptr(gx), stream_ptr(gy.device)))


# ------------------------------------------------------------------------------------------------
# slice-list convolutions (csrc/conv_seg.hip): the shape-changing units' convs as sums of stride-1 tap-subset convs

class ConvSeg:
    """A plan of nbdt_conv_seg_*: `classes` = [{"slices": [(tensor, ch0, [halo taps 3R+S], w_matrix, [w_off per tap])],
    "out": (bs, hs, ws, base)}], over input tensors [B][gh+2][gw+2][pix_strides[i]] and weight matrices
    [cout][w_row_strides[j]].  Host-side object; the device tables appear at the first launch."""

    def __init__(self, B, gh, gw, cout, pix_strides, w_row_strides, classes, tile=0, nbuf=0, flops=None):
        d = _C.ConvSegDesc()
        d.B, d.gh, d.gw, d.cout = B, gh, gw, cout
        d.ntensors, d.nmatrices, d.nclasses = len(pix_strides), len(w_row_strides), len(classes)
        _fill(d.pix_stride, pix_strides)
        _fill(d.w_row_stride, w_row_strides)
        d.tile, d.nbuf = tile, nbuf
        self._keep = []
        for c, k in enumerate(classes):
            arr = (_C.ConvSegSlice * len(k["slices"]))()
            for s, (tensor, ch0, taps, wm, woffs) in enumerate(k["slices"]):
                arr[s].tensor, arr[s].ch0, arr[s].ntaps, arr[s].w_matrix = tensor, ch0, len(taps), wm
                _fill(arr[s].tap, taps)
                _fill(arr[s].w_off, woffs)
            self._keep.append(arr)
            d.cls[c].nslices = len(k["slices"])
            d.cls[c].slices = ctypes.cast(arr, ctypes.POINTER(_C.ConvSegSlice))
            d.cls[c].out_bs, d.cls[c].out_hs, d.cls[c].out_ws, d.cls[c].out_base = k["out"]
        self.desc, self.classes = d, classes
        self.handle = ctypes.c_void_p()
        check(lib().nbdt_conv_seg_create(ctypes.byref(d), ctypes.byref(self.handle)))
        tile_, nbuf_, rounds = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        steps = (ctypes.c_int32 * 4)()
        wel = ctypes.c_int64()
        check(lib().nbdt_conv_seg_info(self.handle, ctypes.byref(tile_), ctypes.byref(nbuf_), steps, ctypes.byref(rounds),
                                       ctypes.byref(wel)))
        self.tile, self.nbuf, self.max_rounds, self.w_tile_elems = tile_.value, nbuf_.value, rounds.value, wel.value
        self.nsteps = list(steps)[:len(classes)]
        # algorithmic flops of a launch (the launch timers): 2 x pixels x cout x 32 per K step unless the caller knows better
        self.flops = float(flops) if flops is not None else 2.0 * B * gh * gw * cout * 32 * sum(self.nsteps)

    def steps(self, cls):
        """(records [n][8] int32, prologue slices) of class `cls` -- host copy, no device work."""
        import numpy as np
        n = self.nsteps[cls]
        buf = (ctypes.c_int32 * (8 * n))()
        npro = ctypes.c_int32()
        rc = lib().nbdt_conv_seg_steps(self.handle, cls, buf, n, ctypes.byref(npro))
        if rc < 0:
            check(rc)
        return np.frombuffer(buf, dtype=np.int32).reshape(n, 8).copy(), npro.value

    def tile_weights(self, ws, out=None):
        """ws: the bf16 weight matrices -> the DMA-ordered tile buffer the launches read."""
        dev = ws[0].device
        if out is None:
            out = torch.empty(self.w_tile_elems, dtype=torch.bfloat16, device=dev)
        arr = (ctypes.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
        check(lib().nbdt_conv_seg_tile_weights(self.handle, arr, ptr(out), stream_ptr(dev)))
        return out

    def __call__(self, ins, w, out, residual=None, bn_scratch=None):
        """w: the tile buffer (bf16 path) or, for fp32 tensors (verification-only reference mode), the list of fp32
        weight matrices in their plain layout."""
        dev = ins[0].device
        arr = (ctypes.c_void_p * len(ins))(*[t.data_ptr() for t in ins])
        if _ref(ins[0]):
            warr = (ctypes.c_void_p * len(w))(*[t.data_ptr() for t in w])
            check(lib().nbdt_ref_conv_seg(self.handle, arr, warr, ptr(out), ptr(residual), stream_ptr(dev)))
            if bn_scratch is not None:
                B, H, W, C = _dims(out)
                check(lib().nbdt_ref_bn_stats(ptr(out), B, H, W, C, BN_EPS, BN_MOMENTUM, None, None, None, None,
                                              ptr(bn_scratch), stream_ptr(dev)))
            return
        ev = None
        if _timer is not None and _timer.wants("conv_igemm"):
            ev = _timer.bracket("conv_igemm", self.flops, dev)
            ev[0].record()
        check(lib().nbdt_conv_seg(self.handle, arr, ptr(w), ptr(out), ptr(residual), ptr(bn_scratch), stream_ptr(dev)))
        if ev is not None:
            ev[1].record()
            _timer.count(last_igemm_kernel_full())

    def __del__(self):
        try:
            h = getattr(self, "handle", None)
            if h is not None and h.value:
                lib().nbdt_conv_seg_destroy(h)
                self.handle = None
        except Exception:      # (interpreter shutdown: modules may already be gone)
            pass


def _plain_out(H, W, C):
    row = (W + 2) * C
    return ((H + 2) * row, row, C, row + C)


# 1-D phase tables of a stride-2 3x3 / pad-1 conv.  Forward over the space-to-depth input: phase p (row parity of the
# input) holds kernel rows r with halo row R (input row y + R - 1 of the half-resolution grid): out(y) reads
# in(2y + r - 1).  Data gradient: output parity p (of the unpadded input row) gets kernel rows r from gradient row u + R - 1.
S2_FWD = {0: [(1, 1)], 1: [(0, 0), (2, 1)]}
S2_BWD = {0: [(1, 1)], 1: [(0, 2), (2, 1)]}


def _interleave(major, minor):
    """`minor` slices spread evenly among `major` ones (never first: the first slice is every tile's prologue)."""
    if not minor:
        return list(major)
    out, j = [], 0
    for i, s in enumerate(major):
        out.append(s)
        while j < len(minor) and (j + 1) * len(major) <= (i + 1) * len(minor):
            out.append(minor[j])
            j += 1
    out.extend(minor[j:])
    return out


def seg_fwd_s2(B, Hi, Wi, cin, cout, tile=0, nbuf=0, order=(3, 0, 1, 2)):
    """Conv2d(3x3, stride 2, pad 1) forward over the space-to-depth copy [B][Hi/2+2][Wi/2+2][4 cin] of its input
    (bn_apply_s2d); weight matrix 0 = the forward weights [cout][9][cin].  order: phases (2 p + q) per 32-channel chunk."""
    Ho, Wo = Hi // 2, Wi // 2
    slices = []
    for kc in range(cin // 32):
        for ph in order:
            p, q = ph >> 1, ph & 1
            taps = [(3 * R + S, 3 * r + s) for (r, R) in S2_FWD[p] for (s, S) in S2_FWD[q]]
            slices.append((0, ph * cin + kc * 32, [t for t, _ in taps], 0, [w * cin + kc * 32 for _, w in taps]))
    return ConvSeg(B, Ho, Wo, cout, [4 * cin], [9 * cin], [{"slices": slices, "out": _plain_out(Ho, Wo, cout)}],
                   tile=tile, nbuf=nbuf, flops=2.0 * B * Ho * Wo * 9 * cin * cout)


def seg_conv3x3_plus_1x1(B, H, W, cin, cout, cin_sc, sc_pix_stride, tile=0, nbuf=0):
    """conv3x3(stride 1)(x) + conv1x1(s): tensor 0 = x [..][cin], tensor 1 = s with `sc_pix_stride` channels per pixel of
    which the first cin_sc are the shortcut's input (the plain tensor, or phase (0,0) of a space-to-depth copy: a 1x1
    stride-2 conv); matrices: [cout][9][cin], [cout][cin_sc]."""
    major = [(0, kc * 32, list(range(9)), 0, [t * cin + kc * 32 for t in range(9)]) for kc in range(cin // 32)]
    minor = [(1, kc * 32, [4], 1, [kc * 32]) for kc in range(cin_sc // 32)]
    return ConvSeg(B, H, W, cout, [cin, sc_pix_stride], [9 * cin, cin_sc],
                   [{"slices": _interleave(major, minor), "out": _plain_out(H, W, cout)}], tile=tile, nbuf=nbuf,
                   flops=2.0 * B * H * W * cout * (9 * cin + cin_sc))


def seg_dgrad_s2(B, Hi, Wi, cin, cout, shortcut=False, tile=0, nbuf=0):
    """Data gradient of Conv2d(cin -> cout, 3x3, stride 2, pad 1) [+ of the 1x1 stride-2 shortcut next to it]: tensor 0 =
    dL/d(conv output) [B][Hi/2+2][Wi/2+2][cout] (tensor 1 = dL/d(shortcut output), same shape); output dL/d(input)
    [B][Hi+2][Wi+2][cin], every interior pixel written once (four parity classes); matrices: the transposed tap-reversed
    copy wd [cin][9][cout] (weight_prep), and the shortcut's [cin][cout]."""
    Ho, Wo = Hi // 2, Wi // 2
    rowx = (Wi + 2) * cin
    classes = []
    for ph in (1, 0):
        for pw in (1, 0):
            taps = [(3 * R + S, 8 - (3 * r + s)) for (r, R) in S2_BWD[ph] for (s, S) in S2_BWD[pw]]
            major = [(0, kc * 32, [t for t, _ in taps], 0, [w * cout + kc * 32 for _, w in taps])
                     for kc in range(cout // 32)]
            minor = [(1, kc * 32, [4], 1, [kc * 32]) for kc in range(cout // 32)] if (shortcut and ph == 0 and pw == 0) else []
            classes.append({"slices": _interleave(major, minor),
                            "out": ((Hi + 2) * rowx, 2 * rowx, 2 * cin, (ph + 1) * rowx + (pw + 1) * cin)})
    return ConvSeg(B, Ho, Wo, cin, [cout, cout] if shortcut else [cout], [9 * cout, cout] if shortcut else [9 * cout],
                   classes, tile=tile, nbuf=nbuf, flops=2.0 * B * Ho * Wo * cin * cout * (9 + (1 if shortcut else 0)))


def seg_dgrad3x3_plus_1x1(B, H, W, cin, cout, tile=0, nbuf=0):
    """Data gradient of conv3x3(cin -> cout, stride 1) plus that of the 1x1 shortcut beside it (stride 1): tensors
    dL/d(conv output), dL/d(shortcut output) [B][H+2][W+2][cout]; matrices wd [cin][9][cout], wd_sc [cin][cout]."""
    major = [(0, kc * 32, list(range(9)), 0, [t * cout + kc * 32 for t in range(9)]) for kc in range(cout // 32)]
    minor = [(1, kc * 32, [4], 1, [kc * 32]) for kc in range(cout // 32)]
    return ConvSeg(B, H, W, cin, [cout, cout], [9 * cout, cout],
                   [{"slices": _interleave(major, minor), "out": _plain_out(H, W, cin)}], tile=tile, nbuf=nbuf,
                   flops=2.0 * B * H * W * cin * cout * 10)


def bn_apply_s2d(x, mean, rstd, gamma, beta, y, relu=True):
    """y = [relu](bn(x)) as the space-to-depth copy [B][H/2+2][W/2+2][4C] (nbdt_bn_apply_s2d)."""
    B, H, W, C = _dims(x)
    fn = lib().nbdt_ref_bn_apply_s2d if _ref(x) else lib().nbdt_bn_apply_s2d
    check(fn(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), 1 if relu else 0, B, H, W, C, ptr(y),
             stream_ptr(x.device)))


def s2d_buffer(B, H, W, C, device, dtype=torch.bfloat16):
    """Zero-initialised space-to-depth activation buffer [B][H/2+2][W/2+2][4C] for an [H, W, C] tensor."""
    return torch.zeros((B, H // 2 + 2, W // 2 + 2, 4 * C), dtype=dtype, device=device)


def conv_wgrad_desc_s2d(B, Hi, Wi, cin, cout, k):
    """Weight gradient of Conv2d(k in {1,3}, stride 2, pad k//2) when its input is stored as the space-to-depth copy
    [B][Hi/2+2][Wi/2+2][4 cin] (bn_apply_s2d): tap (r, s) reads phase (r != 1, s != 1) at half-resolution pixel
    (y + (r == 0 ? -1 : 0), x + ...) -- every tap a unit-stride pixel walk of 4 cin-element pixels."""
    Ho, Wo = Hi // 2, Wi // 2
    d = WgradDesc()
    d.B, d.gh, d.gw, d.cin, d.cout = B, Ho, Wo, cin, cout
    row2 = (Wo + 2) * 4 * cin
    if k == 3:
        d.ntaps = d.w_ntaps = 9
        offs = []
        for r in range(3):
            for s in range(3):
                p, q = int(r != 1), int(s != 1)
                dy, dx = (-1 if r == 0 else 0), (-1 if s == 0 else 0)
                offs.append((1 + dy) * row2 + (1 + dx) * 4 * cin + (2 * p + q) * cin)
        _fill(d.tap_off, offs)
        _fill(d.w_tap, range(9))
        d.x_base = 0
    else:
        d.ntaps = d.w_ntaps = 1
        d.tap_off[0] = 0
        d.w_tap[0] = 0
        d.x_base = row2 + 4 * cin
    d.x_bs, d.x_hs, d.x_ws = (Ho + 2) * row2, row2, 4 * cin
    rowo = (Wo + 2) * cout
    d.g_bs, d.g_hs, d.g_ws, d.g_base = (Ho + 2) * rowo, rowo, cout, rowo + cout
    return d


def conv_fwd_desc_s2d_1x1(B, Hi, Wi, cin, cout):
    """Conv2d(1x1, stride 2) forward over the space-to-depth copy of its input: phase (0, 0), unit-stride pixels."""
    Ho, Wo = Hi // 2, Wi // 2
    d = ConvDesc()
    d.B, d.gh, d.gw, d.cin, d.cout = B, Ho, Wo, cin, cout
    row2 = (Wo + 2) * 4 * cin
    d.ntaps = d.w_ntaps = 1
    d.tap_off[0] = 0
    d.w_tap[0] = 0
    d.in_bs, d.in_hs, d.in_ws, d.in_base = (Ho + 2) * row2, row2, 4 * cin, row2 + 4 * cin
    rowo = (Wo + 2) * cout
    d.out_bs, d.out_hs, d.out_ws, d.out_base = (Ho + 2) * rowo, rowo, cout, rowo + cout
    d.accumulate = 0
    d.wide_tile = 1
    return d
