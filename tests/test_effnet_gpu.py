"""EfficientNet-B0 path: MBConv kernels against plain fp32 PyTorch ops on the
same bf16-rounded inputs, then the whole engine against the fp32 oracle restatement
(oracle/torch_models.py::EfficientNetB0) with identical weights.

Tolerances: bf16 outputs within 2^-7 relative (one bf16 ulp = 2^-8) of the fp32 result computed from
the same bf16 inputs; fp32 reductions 2e-3 relative; end-to-end the same direction/norm criteria as
the WRN engine (bf16 storage, see test_engine_gpu.py)."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import nbdt_oracle as O
import torch_models as TM

pytestmark = pytest.mark.gpu

from nbdt import ops  # noqa: E402
from nbdt.engine import train_step  # noqa: E402
from nbdt.engine_effnet import EfficientNetEngine  # noqa: E402
from nbdt.loss import SoftTreeSupLoss  # noqa: E402

DEV = "cuda:0"


def _padded_from(x_nchw):
    """fp32 NCHW (already bf16-representable) -> padded NHWC bf16 device buffer."""
    B, C, H, W = x_nchw.shape
    t = ops.padded(B, H, W, C, DEV)
    ops.interior(t).copy_(x_nchw.permute(0, 2, 3, 1).to(DEV))
    return t


def _nchw(t):
    return ops.interior(t).float().permute(0, 3, 1, 2).cpu()


def _bf(x):
    return x.to(torch.bfloat16).float()


def _close_bf16(got, want, what, rel=2 ** -7, abs_=1e-3):
    err = (got - want).abs()
    tol = rel * want.abs() + abs_
    assert (err <= tol).all(), f"{what}: max err {err.max().item():.4g} at |ref| {want.abs().max().item():.4g}"


def _swish(v):
    return v * torch.sigmoid(v)


@pytest.mark.parametrize("k,stride,C,H,W", [(3, 1, 32, 8, 8), (3, 2, 96, 12, 16), (5, 1, 160, 6, 10),
                                             (5, 2, 64, 16, 8), (3, 1, 672, 4, 4), (5, 2, 1152, 4, 4)])
def test_depthwise_conv_forward_and_gradients(k, stride, C, H, W):
    g = torch.Generator().manual_seed(k * 100 + C)
    B = 3
    x = _bf(torch.randn(B, C, H, W, generator=g))
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    gy = _bf(torch.randn(B, C, H // stride, W // stride, generator=g))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, stride, k // 2, 1, C)
    y_ref.backward(gy)

    wt = w.view(C, k * k).t().contiguous().to(DEV)          # [taps][C]
    xp, gyp = _padded_from(x), _padded_from(gy)
    yp = ops.padded(B, H // stride, W // stride, C, DEV)
    ops.dwconv_fwd(xp, wt, yp, k, stride)
    _close_bf16(_nchw(yp), y_ref.detach(), "dw fwd")
    gxp = ops.padded(B, H, W, C, DEV)
    ops.dwconv_bwd_data(gyp, wt, gxp, k, stride)
    _close_bf16(_nchw(gxp), xr.grad, "dw bwd_data")
    dw = torch.zeros(k * k, C, device=DEV)
    ops.dwconv_bwd_weight(xp, gyp, dw, k, stride)
    ops.dwconv_bwd_weight(xp, gyp, dw, k, stride)           # accumulates (+=)
    want = 2 * wr.grad.view(C, k * k).t()
    assert (dw.cpu() - want).abs().max().item() <= 2e-3 * want.abs().max().item() + 1e-4
    # borders of the outputs stay zero
    assert yp[:, 0].abs().max().item() == 0 and gxp[:, :, 0].abs().max().item() == 0
    # fused batch statistics of the output (what the next BatchNorm consumes)
    scratch = torch.zeros(ops.BN_SLOTS * 2 * 2048, device=DEV)
    mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    yp2 = ops.padded(B, H // stride, W // stride, C, DEV)
    ops.dwconv_fwd(xp, wt, yp2, k, stride, bn_scratch=scratch)
    ops.bn_stats(yp2, scratch, mean, rstd, slots_filled=True)
    assert torch.equal(yp2, yp) and scratch.abs().max().item() == 0
    yq = _nchw(yp)
    want_mean = yq.mean((0, 2, 3))
    want_rstd = 1.0 / torch.sqrt(yq.var((0, 2, 3), unbiased=False) + 1e-5)
    assert (mean.cpu() - want_mean).abs().max().item() < 1e-3 * (1 + want_mean.abs().max().item())
    assert ((rstd.cpu() - want_rstd).abs() / want_rstd).max().item() < 2e-3


@pytest.mark.parametrize("B,k,stride,C,H,W", [(11, 5, 1, 1152, 7, 7),     # 5 channel blocks of 29 chunks (the last short), batch chunks 9 + 2
                                               (5, 3, 1, 96, 56, 56),       # long rows: one pixel in flight, batch chunks 2 + 2 + 1
                                               (5, 3, 2, 144, 56, 56),      # stride 2, 28 outputs per row: 4 + 1
                                               (7, 5, 1, 672, 14, 14),      # 3 channel blocks of 28, rows of 14 in two trips of 8
                                               (4, 3, 1, 40, 9, 13)])       # odd sizes, 5 chunks, one block
def test_depthwise_weight_gradient_over_channel_blocks_and_batch_chunks(B, k, stride, C, H, W):
    """nbdt_dwconv_bwd_weight at the launch shapes EfficientNet-B0's layers produce (channel blocks in blockIdx.x, several
    images per block, 8 / 4 / 1 output pixels' loads in flight): against torch autograd in fp32 on the same bf16 operands,
    in the default and the deterministic (row per block + ordered fold) mode, twice (the second call accumulates)."""
    g = torch.Generator().manual_seed(B * 1000 + C)
    x = _bf(torch.randn(B, C, H, W, generator=g))
    gy = _bf(torch.randn(B, C, H // stride, W // stride, generator=g))
    wr = torch.zeros(C, 1, k, k, requires_grad=True)
    F.conv2d(x, wr, None, stride, k // 2, 1, C)[:, :, :H // stride, :W // stride].backward(gy)
    want = wr.grad.view(C, k * k).t()
    xp, gyp = _padded_from(x), _padded_from(gy)
    for det in (False, True):
        ops.set_deterministic(det)
        try:
            dw = torch.zeros(k * k, C, device=DEV)
            ops.dwconv_bwd_weight(xp, gyp, dw, k, stride)
            one = dw.clone()
            ops.dwconv_bwd_weight(xp, gyp, dw, k, stride)
            if det:
                again = torch.zeros(k * k, C, device=DEV)
                ops.dwconv_bwd_weight(xp, gyp, again, k, stride)
                assert torch.equal(again, one)                      # bit-reproducible
        finally:
            ops.set_deterministic(False)
        scale = want.abs().max().item()
        assert (one.cpu() - want).abs().max().item() <= 1e-4 * scale, (det, (one.cpu() - want).abs().max().item(), scale)
        assert (dw.cpu() - 2 * want).abs().max().item() <= 2e-4 * scale


@pytest.mark.parametrize("C,H,W,act", [(32, 8, 8, ops.ACT_SWISH), (96, 5, 7, ops.ACT_SWISH), (160, 4, 4, ops.ACT_NONE),
                                       (1280, 2, 2, ops.ACT_SWISH), (64, 6, 6, ops.ACT_RELU)])
def test_bn_act_apply_pool_and_backward_forms(C, H, W, act):
    g = torch.Generator().manual_seed(C + H)
    B = 4
    f = {ops.ACT_SWISH: _swish, ops.ACT_RELU: torch.relu, ops.ACT_NONE: lambda v: v}[act]
    x = _bf(torch.randn(B, C, H, W, generator=g) * 2 + 0.3)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    gate = torch.rand(B, C, generator=g)
    res = _bf(torch.randn(B, C, H, W, generator=g))
    gu = _bf(torch.randn(B, C, H, W, generator=g))
    gpool = torch.randn(B, C, generator=g)
    xp = _padded_from(x)
    mean = torch.empty(C, device=DEV)
    rstd = torch.empty(C, device=DEV)
    scratch = torch.zeros(ops.BN_SLOTS * 2 * 2048, device=DEV)
    ops.bn_stats(xp, scratch, mean, rstd)
    gd, bd = gamma.to(DEV), beta.to(DEV)

    def ref_forward(xr, gr, br):
        return f(F.batch_norm(xr, None, None, gr, br, True, 0.0, 1e-5))

    # apply: plain, gated, gated + residual
    yp = ops.padded(B, H, W, C, DEV)
    a_ref = ref_forward(x, gamma, beta)
    ops.bn_act_apply(xp, mean, rstd, gd, bd, yp, act=act)
    _close_bf16(_nchw(yp), a_ref, "apply")
    ops.bn_act_apply(xp, mean, rstd, gd, bd, yp, act=act, gate=gate.to(DEV), residual=_padded_from(res))
    _close_bf16(_nchw(yp), a_ref * gate[:, :, None, None] + res, "apply gate+res", abs_=4e-3)
    # pooled mean, and pooled product with an upstream gradient
    out = torch.full((B, C), 7.0, device=DEV)
    ops.bn_act_pool(xp, mean, rstd, gd, bd, out, act=act)
    assert (out.cpu() - a_ref.mean((2, 3))).abs().max().item() < 2e-3
    ops.bn_act_pool(xp, mean, rstd, gd, bd, out, act=act, mul=_padded_from(gu), scale=1.0)
    want = (a_ref * gu).sum((2, 3))
    assert (out.cpu() - want).abs().max().item() < 2e-3 * max(1.0, want.abs().max().item())

    # backward, three forms, against autograd through batch_norm + act (+ SE scaling / pooling)
    for form in ("plain", "se", "pool"):
        xr = x.clone().requires_grad_(True)
        gr = gamma.clone().requires_grad_(True)
        br = beta.clone().requires_grad_(True)
        a = ref_forward(xr, gr, br)
        if form == "plain":
            a.backward(gu)
            args = dict(gu=_padded_from(gu))
        elif form == "se":
            (a * gate[:, :, None, None]).backward(gu, retain_graph=True)
            a.mean((2, 3)).backward(gpool)
            args = dict(gu=_padded_from(gu), gate=gate.to(DEV), gpool=gpool.to(DEV))
        else:
            a.mean((2, 3)).backward(gpool)
            args = dict(gu=None, gpool=gpool.to(DEV))
        dsum = torch.empty(2 * C, device=DEV)
        dgamma = torch.zeros(C, device=DEV)
        dbeta = torch.zeros(C, device=DEV)
        gx = ops.padded(B, H, W, C, DEV)
        add = _bf(torch.randn(B, C, H, W, generator=g)) if form != "pool" else None
        ops.bn_act_bwd(args["gu"], xp, mean, rstd, gd, bd, scratch, dsum, dgamma, dbeta, gx, act=act,
                       gate=args.get("gate"), gpool=args.get("gpool"),
                       gx_add=_padded_from(add) if add is not None else None)
        want_gx = xr.grad + (add if add is not None else 0)
        scale = want_gx.abs().max().item()
        assert (_nchw(gx) - want_gx).abs().max().item() < 2e-2 * scale + 1e-3, form
        assert (dgamma.cpu() - gr.grad).abs().max().item() < 5e-3 * gr.grad.abs().max().item() + 1e-3, form
        assert (dbeta.cpu() - br.grad).abs().max().item() < 5e-3 * br.grad.abs().max().item() + 1e-3, form
        assert scratch.abs().max().item() == 0, "scratch must be left zeroed"


@pytest.mark.parametrize("B,C,H,W", [(4, 32, 8, 8), (3, 96, 14, 14), (5, 672, 7, 7), (2, 1152, 7, 7), (2, 160, 28, 28)])
def test_se_backward_in_one_reduction_pass(B, C, H, W):
    """nbdt_bn_act_se_sums + nbdt_bn_act_se_bwd_apply against autograd through swish(bn(x)) * gate and the pooled branch,
    and against the three-pass form (nbdt_bn_act_pool(mul) + nbdt_bn_act_bwd) on the same inputs: dL/dgate, dgamma, dbeta
    to fp32 summation noise, the input gradient to one bf16 rounding."""
    g = torch.Generator().manual_seed(B * C + H)
    x = _bf(torch.randn(B, C, H, W, generator=g) * 2 + 0.3)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    gate = torch.rand(B, C, generator=g)
    gu = _bf(torch.randn(B, C, H, W, generator=g))
    gpool = torch.randn(B, C, generator=g)
    xp, gup = _padded_from(x), _padded_from(gu)
    mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    scratch = torch.zeros(ops.BN_SLOTS * 2 * 2048, device=DEV)
    ops.bn_stats(xp, scratch, mean, rstd)
    gd, bd, gate_d, gpool_d = gamma.to(DEV), beta.to(DEV), gate.to(DEV), gpool.to(DEV)
    # autograd reference
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = _swish(F.batch_norm(xr, None, None, gr, br, True, 0.0, 1e-5))
    (a * gate[:, :, None, None]).backward(gu, retain_graph=True)
    a.mean((2, 3)).backward(gpool)
    want_dgate = (a.detach() * gu).sum((2, 3))
    # one-pass form
    sums = torch.zeros((5, B, C), device=DEV)      # zero on entry; the apply call leaves it zero again
    ops.bn_act_se_sums(gup, xp, mean, rstd, gd, bd, sums)
    dgate1 = sums[0].clone()
    assert (dgate1.cpu() - want_dgate).abs().max().item() < 2e-3 * max(1.0, want_dgate.abs().max().item())
    dsum, dgamma, dbeta = torch.empty(2 * C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    gx = ops.padded(B, H, W, C, DEV)
    ops.bn_act_se_bwd_apply(gup, gate_d, gpool_d, sums, xp, mean, rstd, gd, bd, dsum, dgamma, dbeta, gx)
    scale = xr.grad.abs().max().item()
    assert (_nchw(gx) - xr.grad).abs().max().item() < 2e-2 * scale + 1e-3
    assert (dgamma.cpu() - gr.grad).abs().max().item() < 5e-3 * gr.grad.abs().max().item() + 1e-3
    assert (dbeta.cpu() - br.grad).abs().max().item() < 5e-3 * br.grad.abs().max().item() + 1e-3
    assert sums.abs().max().item() == 0, "the sums buffer must be left zeroed"
    # three-pass form on the same inputs
    dgate3 = torch.empty(B, C, device=DEV)
    ops.bn_act_pool(xp, mean, rstd, gd, bd, dgate3, mul=gup, scale=1.0)
    dsum3, dgamma3, dbeta3 = torch.empty(2 * C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    gx3 = ops.padded(B, H, W, C, DEV)
    ops.bn_act_bwd(gup, xp, mean, rstd, gd, bd, scratch, dsum3, dgamma3, dbeta3, gx3, gate=gate_d, gpool=gpool_d)
    assert (dgate1 - dgate3).abs().max().item() < 1e-4 * max(1.0, dgate3.abs().max().item())
    assert (dsum - dsum3).abs().max().item() < 1e-4 * max(1.0, dsum3.abs().max().item())
    assert (gx.float() - gx3.float()).abs().max().item() <= 2 ** -7 * gx3.float().abs().max().item()
    # deterministic mode refuses the atomics form (the engine takes the three-pass form there)
    ops.set_deterministic(True)
    try:
        with pytest.raises(Exception):
            ops.bn_act_se_sums(gup, xp, mean, rstd, gd, bd, sums)
    finally:
        ops.set_deterministic(False)


@pytest.mark.parametrize("C,Cr,S", [(32, 32, 8), (160, 144, 6), (1152, 1152, 48)])
def test_se_gate_forward_backward(C, Cr, S):
    g = torch.Generator().manual_seed(C + S)
    B = 5
    pooled = torch.zeros(B, C)
    pooled[:, :Cr] = torch.randn(B, Cr, generator=g)
    w1 = torch.randn(S, Cr, generator=g) * 0.3
    b1 = torch.randn(S, generator=g) * 0.1
    w2 = torch.randn(Cr, S, generator=g) * 0.3
    b2 = torch.randn(Cr, generator=g) * 0.1
    dgate = torch.randn(B, C, generator=g)
    t = [v.clone().requires_grad_(True) for v in (pooled, w1, b1, w2, b2)]
    pre1 = t[0][:, :Cr] @ t[1].t() + t[2]
    gate_ref = torch.sigmoid(_swish(pre1) @ t[3].t() + t[4])
    gate_ref.backward(dgate[:, :Cr])

    d = lambda v: v.to(DEV).contiguous()
    pre1_d = torch.empty(B, S, device=DEV)
    gate_d = torch.empty(B, C, device=DEV)
    ops.se_gate_fwd(d(pooled), d(w1), d(b1), d(w2), d(b2), pre1_d, gate_d, Cr)
    assert (gate_d.cpu()[:, :Cr] - gate_ref.detach()).abs().max().item() < 1e-5
    assert Cr == C or gate_d[:, Cr:].abs().max().item() == 0
    gpool = torch.empty(B, C, device=DEV)
    grads = [torch.zeros_like(d(v)) for v in (w1, b1, w2, b2)]
    ops.se_gate_bwd(d(dgate), gate_d, pre1_d, d(pooled), d(w1), d(w2), torch.empty(B, Cr, device=DEV),
                    torch.empty(B, S, device=DEV), gpool, grads[0], grads[1], grads[2], grads[3], Cr)
    assert (gpool.cpu()[:, :Cr] - t[0].grad[:, :Cr]).abs().max().item() < 1e-4
    for got, ref in zip(grads, t[1:]):
        assert (got.cpu() - ref.grad).abs().max().item() < 1e-4 * max(1.0, ref.grad.abs().max().item())


def test_dropout_and_strided_stem():
    x = torch.randn(64, 1280, device=DEV)
    mask = torch.empty(64, 1280, dtype=torch.uint8, device=DEV)
    y = torch.empty_like(x)
    ops.dropout_fwd(x, 0.2, 123, mask, y)
    keep = mask.float().mean().item()
    assert abs(keep - 0.8) < 0.01
    assert torch.equal(y, x * mask.float() / 0.8) or (y - x * mask.float() / 0.8).abs().max().item() < 1e-6
    mask2 = torch.empty_like(mask)
    ops.dropout_fwd(x, 0.2, 124, mask2, y)
    assert (mask != mask2).float().mean().item() > 0.2           # a different seed is a different mask
    gx = torch.empty_like(x)
    ops.dropout_bwd(x, 0.2, mask, gx)
    assert (gx - x * mask.float() / 0.8).abs().max().item() < 1e-6

    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, 16, 24, generator=g)
    w = torch.randn(32, 3, 3, 3, generator=g) * 0.2
    ref = F.conv2d(img, w, None, 2, 1)
    wk = w.permute(0, 2, 3, 1).contiguous().to(DEV)               # [co][r][s][ci]
    out = ops.padded(2, 8, 12, 32, DEV)
    ops.stem_conv(img.to(DEV), wk, out, 32, stride=2)
    _close_bf16(_nchw(out), ref, "stem s2")
    gy = _bf(torch.randn(2, 32, 8, 12, generator=g))
    wr = w.clone().requires_grad_(True)
    F.conv2d(img, wr, None, 2, 1).backward(gy)
    dw = torch.zeros(32, 3, 3, 3, device=DEV)
    ops.stem_wgrad(img.to(DEV), _padded_from(gy), dw, 32, stride=2)
    assert (dw.cpu() - wr.grad.permute(0, 2, 3, 1)).abs().max().item() < 2e-3 * wr.grad.abs().max().item()


@pytest.mark.parametrize("B,K,N", [(128, 1280, 1000), (37, 200, 65), (5, 64, 64)])
def test_wide_linear_head(B, K, N):
    g = torch.Generator().manual_seed(B + N)
    x = torch.randn(B, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    gz = torch.randn(B, N, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    zr = F.linear(xr, wr, br)
    zr.backward(gz)
    z = torch.empty(B, N, device=DEV)
    ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), z)
    assert (z.cpu() - zr.detach()).abs().max().item() < 1e-4 * zr.abs().max().item()
    gx = torch.empty(B, K, device=DEV)
    gw = torch.ones(N, K, device=DEV)          # accumulated into (+=)
    gb = torch.ones(N, device=DEV)
    ops.linear_bwd(x.to(DEV), w.to(DEV), gz.to(DEV), gx, gw, gb)
    assert (gx.cpu() - xr.grad).abs().max().item() < 1e-4 * xr.grad.abs().max().item()
    assert (gw.cpu() - 1 - wr.grad).abs().max().item() < 1e-4 * wr.grad.abs().max().item()
    assert (gb.cpu() - 1 - br.grad).abs().max().item() < 1e-4 * br.grad.abs().max().item()


# ---------------------------------------------------------------------------------------- engine

def _rel(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _cos(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def test_state_dict_names_shapes_and_param_count():
    ref = TM.EfficientNetB0(num_classes=1000)
    eng = EfficientNetEngine(num_classes=1000, device=DEV)
    sd_e, sd_r = eng.state_dict(), ref.state_dict()
    assert set(sd_e) == set(sd_r)
    for k in sd_r:
        assert tuple(sd_e[k].shape) == tuple(sd_r[k].shape), k
    n = sum(v.numel() for k, v in eng.named_params("flat").items())
    assert n == 5288548                                   # published EfficientNet-B0 size
    eng.load_state_dict(sd_r)
    back = eng.state_dict()
    for k, v in sd_r.items():
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(back[k].cpu(), v), k


def test_engine_matches_fp32_oracle(pkg_dir):
    torch.manual_seed(0)
    C = 1000
    ref = TM.EfficientNetB0(num_classes=C, dropout_rate=0.0)
    eng = EfficientNetEngine(num_classes=C, dropout_rate=0.0, device=DEV)
    eng.load_state_dict(ref.state_dict())
    otree = O.OracleTree(*O.default_paths("Imagenet1000", "induced-efficientnet_b7b", pkg_dir))
    crit = SoftTreeSupLoss(dataset="Imagenet1000", criterion=nn.CrossEntropyLoss(),
                           hierarchy="induced-efficientnet_b7b")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(16, 3, 96, 96, generator=g)
    y = torch.randint(0, C, (16,), generator=g)
    ref.train()
    z_ref = ref(x)
    loss_ref, dz = O.soft_tree_sup_loss(otree, z_ref.detach().numpy(), y.numpy())
    z_ref.backward(torch.from_numpy(dz))

    eng.zero_grad()
    z = eng.forward(x.to(DEV), training=True)
    loss, gz = crit.loss_and_grad(z, y.to(DEV))
    eng.backward(gz)
    torch.cuda.synchronize()
    scale = z_ref.abs().max().item()
    zerr = (z.cpu() - z_ref.detach()).abs().max().item()
    print(f"logits: max err {zerr:.4f} of scale {scale:.4f}, rel-L2 {_rel(z, z_ref.detach()):.4f}; "
          f"loss {loss.item():.5f} vs {float(loss_ref):.5f}")
    grads = eng.named_params("grad")
    report, bad = [], []
    for name, p in ref.named_parameters():
        c = _cos(grads[name], p.grad)
        ratio = grads[name].float().norm().item() / (p.grad.norm().item() + 1e-30)
        report.append(f"cos {c:.4f} norm-ratio {ratio:.4f} |ref| {p.grad.norm().item():.3e} {name}")
        if p.grad.norm().item() < 1e-6:
            # mathematically zero: the shift of a BatchNorm whose output only feeds 1x1 conv -> BatchNorm
            # (conv3.bn.bias); the oracle holds fp32 cancellation noise, the engine bf16 noise
            if grads[name].float().norm().item() > 1e-3:
                bad.append(report[-1])
        elif not (c > 0.95 and abs(ratio - 1) < 0.15):
            bad.append(report[-1])
    print("\n".join(report))
    # ~80 bf16 storage points between image and logits, each renormalised by a BatchNorm
    assert _rel(z, z_ref.detach()) < 8e-2 and zerr < 0.15 * scale
    assert abs(loss.item() - float(loss_ref)) < 2e-2 * abs(float(loss_ref))
    assert not bad, "\n".join(bad)
    # eval mode uses the running statistics the training step just updated
    ref.eval()
    with torch.no_grad():
        ze_ref = ref(x)
    ze = eng.forward(x.to(DEV), training=False)
    assert (ze.cpu() - ze_ref).abs().max().item() < 5e-2 * ze_ref.abs().max().item()


def test_training_reduces_the_loss_and_full_size_step_runs():
    eng = EfficientNetEngine(num_classes=1000, device=DEV, seed=1)
    crit = SoftTreeSupLoss(dataset="Imagenet1000", criterion=nn.CrossEntropyLoss(),
                           hierarchy="induced-efficientnet_b7b")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16, 3, 224, 224, generator=g).to(DEV)
    y = torch.randint(0, 1000, (16,), generator=g).to(DEV)
    losses = [train_step(eng, crit, x, y, lr=0.05).item() for _ in range(6)]
    assert all(math.isfinite(l) for l in losses), losses
    assert losses[-1] < losses[0], losses


def test_facade_module_is_a_drop_in_backbone_for_nbdt():
    from nbdt.model import HardNBDT, SoftNBDT
    from nbdt.models import efficientnet_b0
    net = efficientnet_b0(num_classes=1000, device=DEV)
    assert "output.fc.weight" in net.state_dict()          # the key reference nbdt/graph.py:393 reads
    x = torch.randn(4, 3, 64, 64, device=DEV)
    soft = SoftNBDT("Imagenet1000", net, hierarchy="induced-efficientnet_b7b")
    hard = HardNBDT("Imagenet1000", net, hierarchy="induced-efficientnet_b7b")
    with torch.no_grad():
        P = soft(x)
        Hh = hard(x)
    assert P.shape == (4, 1000) and abs(P.sum(1) - 1).max().item() < 1e-4
    assert torch.equal(Hh.sum(1), torch.ones(4, device=DEV))


def test_deterministic_mode_covers_the_mbconv_kernels():
    """nbdt_set_deterministic: two engines with one seed, three full training steps each (224x224 would only be
    slower: 64x64 images run every kernel) -- parameters, momentum and running statistics bit-identical, and so are
    two backward passes of one engine.  Without the mode they differ from the first step on (pooled sums, depthwise
    statistics and weight gradients, squeeze-excite parameter gradients all go through fp32 atomics)."""
    from nbdt import ops
    crit = SoftTreeSupLoss(dataset="Imagenet1000", criterion=nn.CrossEntropyLoss(),
                           hierarchy="induced-efficientnet_b7b")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(24, 3, 64, 64, generator=g).to(DEV)
    y = torch.randint(0, 1000, (24,), generator=g).to(DEV)
    ops.set_deterministic(True)
    try:
        runs = []
        for _ in range(2):
            eng = EfficientNetEngine(num_classes=1000, dropout_rate=0.2, device=DEV, seed=4)
            losses = [train_step(eng, crit, x, y, lr=0.05).item() for _ in range(3)]
            torch.cuda.synchronize()
            runs.append((losses, eng.store.flat.clone(), eng.store.mom.clone(),
                         torch.cat([b.running_var for b in eng.bns]).clone()))
        assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
        for a, b in zip(runs[0][1:], runs[1][1:]):
            assert torch.equal(a, b)
        grads = []
        for _ in range(2):
            eng.zero_grad()
            z = eng.forward(x, training=True)
            eng._step -= 1                      # the same dropout mask for both passes
            _, gz = crit.loss_and_grad(z, y)
            eng.backward(gz)
            torch.cuda.synchronize()
            grads.append(eng.store.grad.clone())
        assert torch.equal(grads[0], grads[1])
    finally:
        ops.set_deterministic(False)


def test_two_stream_backward_equals_the_serial_one_bit_for_bit():
    """EfficientNetEngine.backward with the weight gradients on the second stream, the one-unit-lag event wait and
    shared (alternating) gradient buffers, against one stream with private buffers per unit -- deterministic mode, the
    whole flat gradient bit-identical (no launch reads a buffer before its producer finished or after a later unit
    overwrote it)."""
    from nbdt import ops
    crit = SoftTreeSupLoss(dataset="Imagenet1000", criterion=nn.CrossEntropyLoss(),
                           hierarchy="induced-efficientnet_b7b")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 3, 96, 96, generator=g).to(DEV)
    y = torch.randint(0, 1000, (16,), generator=g).to(DEV)
    ops.set_deterministic(True)
    try:
        out = []
        for serial in (False, True, False):
            eng = EfficientNetEngine(num_classes=1000, dropout_rate=0.0, device=DEV, seed=6)
            if serial:
                eng.debug_keep = True
                eng.set_overlap(False)
            eng.zero_grad()
            z = eng.forward(x, training=True)
            loss, gz = crit.loss_and_grad(z, y)
            eng.backward(gz)
            torch.cuda.synchronize()
            out.append((loss.item(), eng.store.grad.clone()))
        assert out[0][0] == out[1][0] == out[2][0]
        assert torch.equal(out[0][1], out[2][1]), "two runs of the two-stream schedule differ"
        rel = ((out[0][1] - out[1][1]).norm() / out[1][1].norm()).item()
        assert torch.equal(out[0][1], out[1][1]), f"two streams + shared buffers vs one stream + private buffers: {rel:.3e}"
    finally:
        ops.set_deterministic(False)


@pytest.mark.parametrize("B,H,W,C,k", [(3, 16, 16, 96, 3), (2, 28, 28, 240, 5), (5, 7, 7, 672, 5), (4, 14, 14, 32, 3)])
def test_depthwise_data_gradient_with_fused_batchnorm_backward_sums(B, H, W, C, k):
    """nbdt_dwconv_bwd_data_bn + nbdt_bn_act_bwd_apply against nbdt_dwconv_bwd_data + nbdt_bn_act_bwd: the same
    data gradient bit for bit, the same BatchNorm backward (sums to fp32 summation order, gx within bf16 rounding of
    them, dgamma / dbeta to 1e-5), in the default and in the deterministic mode."""
    g = torch.Generator().manual_seed(B * 100 + C)
    def act(scale=1.0):
        t = ops.padded(B, H, W, C, DEV)
        ops.interior(t).copy_((torch.randn(B, H, W, C, generator=g) * scale).to(torch.bfloat16).to(DEV))
        return t
    gy, e_raw = act(0.5), act(1.0)
    w = (torch.randn(k * k, C, generator=g) * 0.2).to(DEV)
    mean, rstd = (torch.randn(C, generator=g) * 0.1).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    for det in (False, True):
        ops.set_deterministic(det)
        try:
            scratch = torch.zeros(ops.BN_SLOTS * 2 * 2048, device=DEV)
            ge_ref = ops.padded(B, H, W, C, DEV)
            ops.dwconv_bwd_data(gy, w, ge_ref, k, 1)
            dsum_ref, dg_ref, db_ref = torch.empty(2 * C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            gx_ref = ops.padded(B, H, W, C, DEV)
            ops.bn_act_bwd(ge_ref, e_raw, mean, rstd, gamma, beta, scratch, dsum_ref, dg_ref, db_ref, gx_ref)
            ge = ops.padded(B, H, W, C, DEV)
            ops.dwconv_bwd_data_bn(gy, w, ge, k, e_raw, mean, rstd, gamma, beta, scratch)
            dsum, dg, db = torch.empty(2 * C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            gx = ops.padded(B, H, W, C, DEV)
            ops.bn_act_bwd_apply(ge, e_raw, mean, rstd, gamma, beta, scratch, dsum, dg, db, gx)
            assert torch.equal(ge, ge_ref)
            tol = 2e-5 * dsum_ref.abs().max().item() + 1e-6
            assert (dsum - dsum_ref).abs().max().item() < tol
            assert (dg - dg_ref).abs().max().item() < tol and (db - db_ref).abs().max().item() < tol
            assert (gx.float() - gx_ref.float()).abs().max().item() <= 2 ** -7 * gx_ref.float().abs().max().item()
            assert float(scratch.abs().max()) == 0.0          # slots left zeroed for the next user
        finally:
            ops.set_deterministic(False)


def test_an_interrupted_backward_does_not_poison_the_next_steps_se_sums():
    """The one-pass SE backward accumulates into a buffer that is zero on entry and re-zeroed by its last reader.  A backward
    that stops between the two calls leaves it dirty: the engine notices (host-side flag) and zeroes it before the next use,
    so the next step's gradients equal those of an engine that was never interrupted."""
    crit = SoftTreeSupLoss(dataset="Imagenet1000", criterion=nn.CrossEntropyLoss(), hierarchy="induced-efficientnet_b7b")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    y = torch.randint(0, 1000, (4,), generator=g).to(DEV)
    clean = EfficientNetEngine(num_classes=1000, dropout_rate=0.0, device=DEV, seed=3)
    hurt = EfficientNetEngine(num_classes=1000, dropout_rate=0.0, device=DEV, seed=3)
    real = ops.bn_act_se_bwd_apply
    calls = {"n": 0}

    def failing(*a, **k):
        calls["n"] += 1
        if calls["n"] == 3:
            raise RuntimeError("interrupted")
        return real(*a, **k)

    ops.bn_act_se_bwd_apply = failing
    try:
        hurt.zero_grad()
        z = hurt.forward(x, training=True)
        _, gz = crit.loss_and_grad(z, y)
        with pytest.raises(RuntimeError, match="interrupted"):
            hurt.backward(gz)
    finally:
        ops.bn_act_se_bwd_apply = real
    torch.cuda.synchronize()
    hurt.load_state_dict(clean.state_dict())          # same weights and running statistics again
    outs = []
    for eng in (clean, hurt):
        eng.zero_grad()
        z = eng.forward(x, training=True)
        _, gz = crit.loss_and_grad(z, y)
        eng.backward(gz)
        torch.cuda.synchronize()
        outs.append(eng.store.grad.clone())
    rel = ((outs[0] - outs[1]).norm() / outs[0].norm()).item()
    # fp32 atomics: the summation order differs run to run, and a sum that lands on the other side of a bf16 rounding moves
    # the gradients behind it by an ulp -- 1e-5 ... 1e-3 between two clean engines; the poisoned sums gave 0.31
    assert rel < 2e-2, rel
