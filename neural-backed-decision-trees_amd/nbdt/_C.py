"""ctypes binding of libnbdt_hip.so (include/nbdt_hip.h).

There is NO CPU fallback: if the shared library is missing, or a call is made without a HIP
device, the product path raises.  (The CPU oracle under ``oracle/`` is test infrastructure and
is never imported from here.)
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import numpy as np
import torch

_LIBPATH = os.environ.get("NBDT_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib",
                                                         "libnbdt_hip.so")   # env: kernel A/B builds only
_lib = None

NBDT_F32, NBDT_BF16, NBDT_F16 = 0, 1, 2
_ZTYPE = {torch.float32: NBDT_F32, torch.bfloat16: NBDT_BF16, torch.float16: NBDT_F16}


class NBDTHipError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("B", c_int32), ("gh", c_int32), ("gw", c_int32),
        ("cin", c_int32), ("cout", c_int32),
        ("ntaps", c_int32),
        ("tap_off", c_int32 * 9), ("w_tap", c_int32 * 9), ("w_ntaps", c_int32),
        ("in_bs", c_int32), ("in_hs", c_int32), ("in_ws", c_int32), ("in_base", c_int32),
        ("out_bs", c_int32), ("out_hs", c_int32), ("out_ws", c_int32), ("out_base", c_int32),
        ("accumulate", c_int32),
        ("wide_tile", c_int32),
        ("ksplit", c_int32),
        ("w_tiled", ctypes.c_uint64),
    ]


class WgradDesc(ctypes.Structure):
    _fields_ = [
        ("B", c_int32), ("gh", c_int32), ("gw", c_int32),
        ("cin", c_int32), ("cout", c_int32),
        ("ntaps", c_int32),
        ("tap_off", c_int32 * 9), ("w_tap", c_int32 * 9), ("w_ntaps", c_int32),
        ("x_bs", c_int32), ("x_hs", c_int32), ("x_ws", c_int32), ("x_base", c_int32),
        ("g_bs", c_int32), ("g_hs", c_int32), ("g_ws", c_int32), ("g_base", c_int32),
        ("variant", c_int32),
        ("cu_budget", c_int32),
    ]


class ConvSegSlice(ctypes.Structure):
    _fields_ = [("tensor", c_int32), ("ch0", c_int32), ("ntaps", c_int32), ("tap", c_int32 * 9),
                ("w_matrix", c_int32), ("w_off", c_int32 * 9)]


class ConvSegClass(ctypes.Structure):
    _fields_ = [("nslices", c_int32), ("slices", POINTER(ConvSegSlice)),
                ("out_bs", c_int32), ("out_hs", c_int32), ("out_ws", c_int32), ("out_base", c_int32)]


class ConvSegDesc(ctypes.Structure):
    _fields_ = [("B", c_int32), ("gh", c_int32), ("gw", c_int32), ("cout", c_int32),
                ("ntensors", c_int32), ("pix_stride", c_int32 * 4),
                ("nmatrices", c_int32), ("w_row_stride", c_int32 * 4),
                ("nclasses", c_int32), ("cls", ConvSegClass * 4),
                ("tile", c_int32), ("nbuf", c_int32)]


_P = c_void_p
_I32P = POINTER(c_int32)

# name -> (restype, argtypes): every symbol include/nbdt_hip.h declares
SIGNATURES = {
    "nbdt_last_error": (c_char_p, []),
    "nbdt_version": (c_int, []),
    "nbdt_device_count": (c_int, []),
    "nbdt_set_deterministic": (c_int, [c_int32]),
    "nbdt_get_deterministic": (c_int, []),
    "nbdt_set_wgrad_store_epilogue": (c_int, [c_int32]),
    "nbdt_get_wgrad_store_epilogue": (c_int, []),
    "nbdt_set_reserved_cus": (c_int, [c_int32]),
    "nbdt_probe_mfma_stream": (c_int, [c_int32, c_int32, _P, _P]),
    "nbdt_probe_lds_mfma": (c_int, [c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_get_reserved_cus": (c_int, []),
    "nbdt_tree_create": (c_int, [c_int, c_int, c_int, c_int, _I32P, _I32P, _I32P, _I32P, _I32P, _I32P,
                                 POINTER(c_void_p)]),
    "nbdt_tree_destroy": (c_int, [c_void_p]),
    "nbdt_tree_max_depth": (c_int, [c_void_p]),
    "nbdt_debug_last_igemm": (c_char_p, []),
    "nbdt_debug_last_igemm_full": (c_char_p, []),
    "nbdt_debug_last_wgrad": (c_char_p, []),
    "nbdt_conv_wgrad_blocks": (c_int, [_P]),
    "nbdt_conv_plan": (c_int, [_P, _P, _P]),
    "nbdt_soft_forward": (c_int, [c_void_p, _P, c_int, c_int64, c_int64, _P, _P]),
    "nbdt_soft_backward": (c_int, [c_void_p, _P, c_int, c_int64, c_int64, _P, _P, _P]),
    "nbdt_soft_tree_loss": (c_int, [c_void_p, _P, c_int, c_int64, c_int64, _P, c_float, c_float, c_float,
                                    _P, _P, _P, _P]),
    "nbdt_head_soft_tree_loss": (c_int, [c_void_p, _P, _P, _P, c_int64, c_int32, _P, c_float, c_float, c_float,
                                         _P, _P, _P, _P, _P, _P, _P]),
    "nbdt_hard_tree_loss": (c_int, [c_void_p, _P, c_int, c_int64, c_int64, _P, c_float, c_float, c_float,
                                    _P, _P, _P, _P]),
    "nbdt_node_logits_backward": (c_int, [c_void_p, _P, c_int64, _P, _P]),
    "nbdt_hard_forward": (c_int, [c_void_p, _P, c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "nbdt_node_outputs": (c_int, [c_void_p, _P, c_int, c_int64, c_int64, _P, _P, _P, _P, _P]),
    "nbdt_conv_igemm": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "nbdt_conv_igemm_multi": (c_int, [_P, c_int32, _P, _P, _P, _P]),
    "nbdt_conv_igemm_stats": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    "nbdt_conv_seg_create": (c_int, [POINTER(ConvSegDesc), POINTER(c_void_p)]),
    "nbdt_conv_seg_destroy": (c_int, [c_void_p]),
    "nbdt_conv_seg_info": (c_int, [c_void_p, _I32P, _I32P, _I32P, _I32P, POINTER(c_int64)]),
    "nbdt_conv_seg_steps": (c_int, [c_void_p, c_int32, _I32P, c_int32, _I32P]),
    "nbdt_conv_seg_tile_weights": (c_int, [c_void_p, POINTER(c_void_p), _P, _P]),
    "nbdt_conv_seg": (c_int, [c_void_p, POINTER(c_void_p), _P, _P, _P, _P, _P]),
    "nbdt_ref_conv_seg": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), _P, _P, _P]),
    "nbdt_bn_apply_s2d": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_ref_bn_apply_s2d": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_bn_finalize": (c_int, [c_int32, c_int32, c_int32, c_int32, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "nbdt_conv_igemm_bnbwd": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nbdt_conv_igemm_affine": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, c_int32, _P]),
    "nbdt_bn_bwd_fold": (c_int, [c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "nbdt_bn_bwd_apply_cus": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, c_int32, _P]),
    "nbdt_bn_bwd_reduce_cus": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, c_int32, _P]),
    "nbdt_bn_bwd_cus": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P,
                                c_int32, _P]),
    "nbdt_conv_wgrad": (c_int, [POINTER(WgradDesc), _P, _P, _P, _P]),
    "nbdt_ref_conv": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "nbdt_ref_wgrad": (c_int, [POINTER(WgradDesc), _P, _P, _P, _P]),
    "nbdt_ref_bn_stats": (c_int, [_P, c_int32, c_int32, c_int32, c_int32, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "nbdt_ref_bn_apply": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_ref_bn_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int32, c_int32, c_int32, c_int32,
                                c_int32, _P, _P, _P, _P, _P, _P]),
    "nbdt_ref_bn_relu_pool": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_ref_stem_conv": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_ref_stem_wgrad": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_ref_bn_act_apply": (c_int, [_P, _P, _P, _P, _P, c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_ref_bn_act_pool": (c_int, [_P, _P, _P, _P, _P, c_int32, _P, c_float, c_int32, c_int32, c_int32, c_int32, _P,
                                     _P]),
    "nbdt_ref_bn_act_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int32, c_int32, c_int32, c_int32, _P,
                                    _P, _P, _P, _P]),
    "nbdt_ref_dwconv_fwd": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_ref_dwconv_bwd_data": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_ref_dwconv_bwd_weight": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_weight_prep": (c_int, [_P, c_int32, c_int32, c_int32, _P, _P, _P]),
    "nbdt_weight_prep_batched": (c_int, [_P, _P, c_int32, c_int64, _P, _P]),
    "nbdt_weight_tile_batched": (c_int, [_P, _P, c_int32, c_int64, _P, _P]),
    "nbdt_bn_stats": (c_int, [_P, c_int32, c_int32, c_int32, c_int32, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "nbdt_bn_apply": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_bn_bwd_reduce": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32,
                                   _P, _P, _P, _P, _P]),
    "nbdt_bn_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32,
                                  c_int32, _P, _P, _P]),
    "nbdt_bn_relu_pool": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_pool_bn_bwd_reduce": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P,
                                        _P, _P, _P]),
    "nbdt_pool_bn_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P,
                                       _P]),
    "nbdt_stem_conv": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_stem_wgrad": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_bn_act_apply": (c_int, [_P, _P, _P, _P, _P, c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_bn_act_pool": (c_int, [_P, _P, _P, _P, _P, c_int32, _P, c_float, c_int32, c_int32, c_int32, c_int32,
                                 _P, _P]),
    "nbdt_bn_act_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int32, c_int32, c_int32, c_int32,
                                _P, _P, _P, _P, _P, _P]),
    "nbdt_dwconv_fwd": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, _P]),
    "nbdt_dwconv_bwd_data": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_dwconv_bwd_data_bn": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nbdt_bn_act_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, _P, c_int32, c_int32, c_int32, c_int32,
                                      _P, _P, _P, _P, _P, _P]),
    "nbdt_bn_act_se_sums": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_bn_act_se_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         _P, _P, _P, _P, _P]),
    "nbdt_dwconv_bwd_weight": (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_se_gate_fwd": (c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P]),
    "nbdt_se_gate_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P,
                                 _P, _P, _P]),
    "nbdt_se_param_grad": (c_int, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "nbdt_dropout_fwd": (c_int, [_P, c_int64, c_float, ctypes.c_uint32, _P, _P, _P]),
    "nbdt_dropout_bwd": (c_int, [_P, c_int64, c_float, _P, _P, _P]),
    "nbdt_linear_fwd": (c_int, [_P, _P, _P, c_int32, c_int32, c_int32, _P, _P]),
    "nbdt_linear_bwd": (c_int, [_P, _P, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P]),
    "nbdt_sgd_step": (c_int, [_P, _P, _P, c_int64, c_float, c_float, c_float, c_float, _P, c_int32, _P]),
}


def _missing(name):
    def raiser(*_a, **_k):
        raise NBDTHipError(f"{_LIBPATH} does not export {name}: stale build, re-run build()")
    return raiser


def exported_symbols():
    """Names from SIGNATURES the loaded library really exports (used by the C-ABI test)."""
    l = ctypes.CDLL(_LIBPATH)
    return [n for n in SIGNATURES if hasattr(l, n)]


def lib():
    """Load the shared library (once).  Raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise NBDTHipError(
                f"{_LIBPATH} is missing: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the NBDT hot path.")
        l = ctypes.CDLL(_LIBPATH)
        # a timing-experiment build (csrc/common.h: -DNBDT_TIMING_BUILD + switches that skip or fake part of a kernel)
        # computes wrong gradients by design: only the measurement scripts under scratch/ may load one
        if hasattr(l, "nbdt_timing_build") and os.environ.get("NBDT_ALLOW_TIMING_BUILD") != "1":
            raise NBDTHipError(f"{_LIBPATH} is a timing-experiment build (exports nbdt_timing_build): its kernels skip "
                               "work on purpose.  Set NBDT_ALLOW_TIMING_BUILD=1 in a measurement script; never train with it.")
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                setattr(l, name, _missing(name))  # stale .so: fail loudly on first use
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def libpath():
    return _LIBPATH


def check(rc):
    if rc != 0:
        raise NBDTHipError(f"libnbdt_hip error {rc}: {lib().nbdt_last_error().decode()}")


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", lambda idx: torch.cuda.current_stream(idx).cuda_stream)


def stream_of(t):
    return c_void_p(_raw_stream(t.device.index))      # = torch.cuda.current_stream(t.device).cuda_stream, without the Stream object


def require_gpu(t, what):
    if not t.is_cuda:
        raise NBDTHipError(
            f"{what}: tensor is on {t.device}; the NBDT hot path runs on MI355X only "
            "(no CPU fallback -- move inputs to a HIP device)")


def ztype_of(t):
    try:
        return _ZTYPE[t.dtype]
    except KeyError:
        raise NBDTHipError(f"unsupported logits dtype {t.dtype}") from None


class TreeHandle:
    """Owns one nbdt_tree on one device."""

    def __init__(self, flat, device_index):
        self.flat = flat
        self.device_index = device_index
        out = c_void_p()
        as_p = lambda a: a.ctypes.data_as(_I32P)
        check(lib().nbdt_tree_create(
            int(device_index), flat.num_classes, flat.num_inodes, flat.root,
            as_p(flat.node_off), as_p(flat.slot_off), as_p(flat.slot_cls),
            as_p(flat.cls_off), as_p(flat.cls_slot), as_p(flat.slot_next), ctypes.byref(out)))
        self.h = out
        self.max_depth = lib().nbdt_tree_max_depth(self.h)

    def __del__(self):
        try:
            if getattr(self, "h", None) and _lib is not None:
                _lib.nbdt_tree_destroy(self.h)
        except Exception:
            pass


def _rows(z, handle):
    """[B, C] tensor with unit column stride -> (tensor, B, ldz).  C must be the hierarchy's class count: the
    kernels index columns by class, so wider logits would be silently truncated and narrower ones over-read."""
    if z.dim() != 2:
        raise NBDTHipError(f"rules layer expects [B, C] logits, got shape {tuple(z.shape)}")
    if z.shape[1] != handle.flat.num_classes:
        raise NBDTHipError(f"logits have {z.shape[1]} columns but the hierarchy has {handle.flat.num_classes} classes")
    if z.stride(1) != 1 or (z.shape[0] > 1 and z.stride(0) < z.shape[1]):
        z = z.contiguous()
    return z, z.shape[0], (z.stride(0) if z.shape[0] > 1 else z.shape[1])


def _class_targets(y, z):
    if y.is_floating_point() or y.dim() != 1:
        raise NBDTHipError("the fused tree losses take class-index targets [B]; soft / probability targets go "
                           "through the composed path (pass a criterion that is not a default CrossEntropyLoss)")
    return y.to(device=z.device, dtype=torch.int64).contiguous()


def soft_forward(handle, z):
    require_gpu(z, "soft_forward")
    handle.flat.require_single_path()
    z, B, ld = _rows(z, handle)
    P = torch.empty((B, handle.flat.num_classes), dtype=torch.float32, device=z.device)
    check(lib().nbdt_soft_forward(handle.h, ptr(z), ztype_of(z), B, ld, ptr(P), stream_of(z)))
    return P


def soft_backward(handle, z, gP):
    require_gpu(z, "soft_backward")
    handle.flat.require_single_path()
    z, B, ld = _rows(z, handle)
    gP = gP.contiguous().float()
    gz = torch.empty((B, handle.flat.num_classes), dtype=torch.float32, device=z.device)
    check(lib().nbdt_soft_backward(handle.h, ptr(z), ztype_of(z), B, ld, ptr(gP), ptr(gz), stream_of(z)))
    return gz


def soft_tree_loss(handle, z, y, w_xent, w_tree, grad_scale=1.0):
    """Returns (loss scalar tensor, gz [B,C] fp32)."""
    require_gpu(z, "soft_tree_loss")
    handle.flat.require_single_path()
    z, B, ld = _rows(z, handle)
    y = _class_targets(y, z)
    row = torch.empty((B,), dtype=torch.float32, device=z.device)
    loss = torch.empty((), dtype=torch.float32, device=z.device)
    gz = torch.empty((B, handle.flat.num_classes), dtype=torch.float32, device=z.device)
    check(lib().nbdt_soft_tree_loss(handle.h, ptr(z), ztype_of(z), B, ld, ptr(y), float(w_xent),
                                    float(w_tree), float(grad_scale), ptr(row), ptr(loss), ptr(gz),
                                    stream_of(z)))
    return loss, gz


def head_soft_tree_loss(handle, pooled, W, bias, y, w_xent, w_tree, grad_scale=1.0, gW=None, gb=None,
                        want_logits=False, want_gpooled=True):
    """Classifier head + SoftTreeSupLoss forward and backward in ONE launch (nbdt_head_soft_tree_loss): returns
    (loss, dL/dpooled or None, logits or None); dL/dW and dL/db are ACCUMULATED into gW / gb when given."""
    require_gpu(pooled, "head_soft_tree_loss")
    handle.flat.require_single_path()
    if pooled.dtype != torch.float32 or W.dtype != torch.float32 or not pooled.is_contiguous() or not W.is_contiguous():
        raise NBDTHipError("head_soft_tree_loss takes contiguous fp32 features [B, K] and weights [C, K]")
    B, K = pooled.shape
    C = handle.flat.num_classes
    if tuple(W.shape) != (C, K):
        raise NBDTHipError(f"classifier weight is {tuple(W.shape)}, the hierarchy has {C} classes over {K} features")
    y = _class_targets(y, pooled)
    row = torch.empty((B,), dtype=torch.float32, device=pooled.device)
    loss = torch.empty((), dtype=torch.float32, device=pooled.device)
    z = torch.empty((B, C), dtype=torch.float32, device=pooled.device) if want_logits else None
    gp = torch.empty((B, K), dtype=torch.float32, device=pooled.device) if want_gpooled else None
    check(lib().nbdt_head_soft_tree_loss(handle.h, ptr(pooled), ptr(W), ptr(bias), B, K, ptr(y), float(w_xent),
                                         float(w_tree), float(grad_scale), ptr(row), ptr(loss), ptr(z), ptr(gp),
                                         ptr(gW), ptr(gb), stream_of(pooled)))
    return loss, gp, z


def hard_tree_loss(handle, z, y, w_xent, w_node, grad_scale=1.0):
    """HardTreeSupLoss fused: returns (loss scalar tensor, gz [B,C] fp32)."""
    require_gpu(z, "hard_tree_loss")
    z, B, ld = _rows(z, handle)
    y = _class_targets(y, z)
    row = torch.empty((B,), dtype=torch.float32, device=z.device)
    loss = torch.empty((), dtype=torch.float32, device=z.device)
    gz = torch.empty((B, handle.flat.num_classes), dtype=torch.float32, device=z.device)
    check(lib().nbdt_hard_tree_loss(handle.h, ptr(z), ztype_of(z), B, ld, ptr(y), float(w_xent),
                                    float(w_node), float(grad_scale), ptr(row), ptr(loss), ptr(gz),
                                    stream_of(z)))
    return loss, gz


def node_logits(handle, z):
    """[B, R] fp32 child logits of every inner node (slot-major)."""
    require_gpu(z, "node_logits")
    z, B, ld = _rows(z, handle)
    logits = torch.empty((B, handle.flat.num_slots), dtype=torch.float32, device=z.device)
    check(lib().nbdt_node_outputs(handle.h, ptr(z), ztype_of(z), B, ld, ptr(logits), None, None, None,
                                  stream_of(z)))
    return logits


def node_logits_backward(handle, gs):
    require_gpu(gs, "node_logits_backward")
    gs = gs.contiguous().float()
    gz = torch.empty((gs.shape[0], handle.flat.num_classes), dtype=torch.float32, device=gs.device)
    check(lib().nbdt_node_logits_backward(handle.h, ptr(gs), gs.shape[0], ptr(gz), stream_of(gs)))
    return gz


def hard_forward(handle, z, want_onehot=True, want_decisions=False):
    require_gpu(z, "hard_forward")
    z, B, ld = _rows(z, handle)
    C, D = handle.flat.num_classes, handle.max_depth
    dev = z.device
    pred = torch.empty((B,), dtype=torch.int64, device=dev)
    onehot = torch.empty((B, C), dtype=torch.float32, device=dev) if want_onehot else None
    if want_decisions:
        pn = torch.empty((B, D), dtype=torch.int32, device=dev)
        pc = torch.empty((B, D), dtype=torch.int32, device=dev)
        pp = torch.empty((B, D), dtype=torch.float32, device=dev)
        pe = torch.empty((B, D), dtype=torch.float32, device=dev)
    else:
        pn = pc = pp = pe = None
    check(lib().nbdt_hard_forward(handle.h, ptr(z), ztype_of(z), B, ld, ptr(pred), ptr(onehot), ptr(pn),
                                  ptr(pc), ptr(pp), ptr(pe), stream_of(z)))
    return pred, onehot, (pn, pc, pp, pe)


def node_outputs(handle, z):
    require_gpu(z, "node_outputs")
    z, B, ld = _rows(z, handle)
    R, N = handle.flat.num_slots, handle.flat.num_inodes
    dev = z.device
    logits = torch.empty((B, R), dtype=torch.float32, device=dev)
    probs = torch.empty((B, R), dtype=torch.float32, device=dev)
    preds = torch.empty((B, N), dtype=torch.int64, device=dev)
    ent = torch.empty((B, N), dtype=torch.float32, device=dev)
    check(lib().nbdt_node_outputs(handle.h, ptr(z), ztype_of(z), B, ld, ptr(logits), ptr(probs),
                                  ptr(preds), ptr(ent), stream_of(z)))
    return logits, probs, preds, ent
