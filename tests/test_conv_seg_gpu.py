"""Slice-list convolutions (csrc/conv_seg.hip) through the C-ABI against plain fp32 PyTorch on the same bf16-rounded
inputs: the stride-2 3x3 forward over the space-to-depth input, conv3x3 + 1x1 shortcut in one launch, the four parity
classes of the strided data gradient with the shortcut's data gradient folded in -- at small shapes that exercise every
tile form (512 / 256 pixels, whole rows / whole images, 2 / 3 halo buffers, ragged last tile), and at the WRN-28-10
shapes of the benched configuration at batch sizes the CPU reference finishes in seconds.

Ops replaced: nbdt/models/resnet.py:56-67; pytorchcv PreResUnit (stride 2) behind nbdt/models/wideresnet.py:1-5."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from nbdt import ops  # noqa: E402

DEV = "cuda:0"


def _rand_act(B, H, W, C, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, H, W, C, generator=g) * scale).to(torch.bfloat16)
    p = ops.padded(B, H, W, C, DEV)
    ops.interior(p).copy_(x.to(DEV))
    return x.float(), p


def _rand_weight(cout, cin, k, seed):
    g = torch.Generator().manual_seed(seed)
    wb = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(torch.bfloat16)
    internal = wb.float().permute(0, 2, 3, 1).reshape(cout, k * k, cin).contiguous()
    return wb.float(), internal.to(torch.bfloat16).to(DEV)


def _close_bf16(got, ref, what):
    got, ref = got.float().cpu(), ref.float().cpu()
    tol = 2.0 ** -7 * ref.abs() + 2e-2 * ref.abs().mean() + 1e-6
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {(got - ref).abs().max():.4g}"


def _border_zero(p):
    t = p.float()
    assert t[:, 0].abs().max() == 0 and t[:, -1].abs().max() == 0
    assert t[:, :, 0].abs().max() == 0 and t[:, :, -1].abs().max() == 0


def _s2d(xp_plain, B, H, W, C):
    """space-to-depth copy of a padded tensor's interior, built with torch (what nbdt_bn_apply_s2d writes)."""
    x = ops.interior(xp_plain)
    y = ops.s2d_buffer(B, H, W, C, DEV)
    yi = ops.interior(y)
    for p in (0, 1):
        for q in (0, 1):
            yi[..., (2 * p + q) * C:(2 * p + q + 1) * C] = x[:, p::2, q::2, :]
    return y


def test_bn_apply_s2d_is_bn_apply_rearranged():
    B, H, W, C = 3, 8, 16, 64
    xf, xp = _rand_act(B, H, W, C, seed=1)
    g = torch.Generator().manual_seed(2)
    mean, rstd = torch.randn(C, generator=g).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    gamma, beta = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    y = ops.padded(B, H, W, C, DEV)
    ops.bn_apply(xp, mean, rstd, gamma, beta, y, relu=True)
    y2 = ops.s2d_buffer(B, H, W, C, DEV)
    ops.bn_apply_s2d(xp, mean, rstd, gamma, beta, y2, relu=True)
    assert torch.equal(y2, _s2d(y, B, H, W, C))
    _border_zero(y2)


FWD_CASES = [   # B, Hi, Wi, cin, cout, tile, nbuf
    (4, 16, 16, 64, 64, 0, 0),       # 8x8 out, NT=2
    (8, 16, 16, 64, 64, 512, 2),     # forced 512-pixel tiles need 3 buffers here: refused below
    (3, 32, 32, 32, 160, 0, 0),      # 16x16 out, NT=5, ragged (768 pixels: one and a half 512 tiles)
    (16, 16, 16, 32, 128, 512, 3),   # 8x8 out, full tiles, NT=4
    (2, 64, 64, 64, 32, 0, 0),       # 32x32 out, NT=1, row tiles
    (5, 8, 8, 96, 160, 256, 3),      # 4x4 out: 16 images per half tile, ragged
]


@pytest.mark.parametrize("B,Hi,Wi,cin,cout,tile,nbuf", FWD_CASES)
def test_strided_forward_over_space_to_depth(B, Hi, Wi, cin, cout, tile, nbuf):
    xf, xp = _rand_act(B, Hi, Wi, cin, seed=1)
    w_oihw, wb = _rand_weight(cout, cin, 3, seed=2)
    Ho, Wo = Hi // 2, Wi // 2
    if (tile, nbuf) == (512, 2):
        with pytest.raises(Exception):
            ops.seg_fwd_s2(B, Hi, Wi, cin, cout, tile=tile, nbuf=nbuf)
        return
    try:
        plan = ops.seg_fwd_s2(B, Hi, Wi, cin, cout, tile=tile, nbuf=nbuf)
    except Exception:
        assert tile == 512 and Ho * Wo * 4 > 160     # (three halo buffers of a 512-pixel tile do not fit)
        pytest.skip("shape does not fit the forced tile")
    xs = _s2d(xp, B, Hi, Wi, cin)
    wt = plan.tile_weights([wb])
    out = ops.padded(B, Ho, Wo, cout, DEV)
    plan([xs], wt, out)
    ref = F.conv2d(xf.permute(0, 3, 1, 2), w_oihw, stride=2, padding=1).permute(0, 2, 3, 1)
    _close_bf16(ops.interior(out), ref, "strided forward")
    _border_zero(out)
    # the same launch with the next BatchNorm's statistics: bit-identical output, partial rows that fold to the sums
    out2 = ops.padded(B, Ho, Wo, cout, DEV)
    M = B * Ho * Wo
    scr = torch.full((((M + 255) // 256) * 2 * cout,), 7.0, device=DEV)
    plan([xs], wt, out2, bn_scratch=scr)
    assert torch.equal(out, out2)
    part = scr.view(-1, 2, cout).sum(0).cpu()
    o = ops.interior(out2).float().cpu().reshape(-1, cout)
    torch.testing.assert_close(part[0], o.sum(0), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(part[1], (o * o).sum(0), rtol=1e-3, atol=1e-2)
    # ... and with a residual
    rf, rp = _rand_act(B, Ho, Wo, cout, seed=3)
    out3 = ops.padded(B, Ho, Wo, cout, DEV)
    plan([xs], wt, out3, residual=rp)
    _close_bf16(ops.interior(out3), ref + rf, "strided forward + residual")


PLUS_CASES = [   # B, H, W (output grid), cin2, cout, cin_sc, strided shortcut
    (4, 8, 8, 64, 64, 32, True),
    (3, 16, 16, 160, 160, 32, False),   # WRN stage-1 unit 1: 16(32) -> 160, plain shortcut input
    (2, 16, 16, 320, 320, 160, True),   # WRN stage 2 unit 1
    (9, 8, 8, 128, 128, 64, True),      # ragged
]


@pytest.mark.parametrize("B,H,W,cin,cout,cin_sc,strided", PLUS_CASES)
def test_conv3x3_plus_shortcut_in_one_launch(B, H, W, cin, cout, cin_sc, strided):
    xf, xp = _rand_act(B, H, W, cin, seed=1)
    w_oihw, wb = _rand_weight(cout, cin, 3, seed=2)
    ws_oihw, wsb = _rand_weight(cout, cin_sc, 1, seed=3)
    if strided:
        sf, sp_plain = _rand_act(B, 2 * H, 2 * W, cin_sc, seed=4)
        sp = _s2d(sp_plain, B, 2 * H, 2 * W, cin_sc)
        sc_ref = F.conv2d(sf.permute(0, 3, 1, 2), ws_oihw, stride=2)
        plan = ops.seg_conv3x3_plus_1x1(B, H, W, cin, cout, cin_sc, 4 * cin_sc)
    else:
        sf, sp = _rand_act(B, H, W, cin_sc, seed=4)
        sc_ref = F.conv2d(sf.permute(0, 3, 1, 2), ws_oihw)
        plan = ops.seg_conv3x3_plus_1x1(B, H, W, cin, cout, cin_sc, cin_sc)
    wt = plan.tile_weights([wb, wsb.view(cout, cin_sc)])
    out = ops.padded(B, H, W, cout, DEV)
    M = B * H * W
    scr = torch.zeros(((M + 255) // 256) * 2 * cout, device=DEV)
    plan([xp, sp], wt, out, bn_scratch=scr)
    ref = (F.conv2d(xf.permute(0, 3, 1, 2), w_oihw, padding=1) + sc_ref).permute(0, 2, 3, 1)
    _close_bf16(ops.interior(out), ref, "conv3x3 + shortcut")
    _border_zero(out)


DGRAD_CASES = [   # B, Hi, Wi, cin, cout, shortcut
    (4, 16, 16, 64, 64, True),
    (4, 16, 16, 64, 64, False),
    (3, 32, 32, 160, 320, True),     # WRN stage 2 unit 1 (16x16 gradient grid, ragged)
    (8, 16, 16, 320, 640, True),     # WRN stage 3 unit 1 (8x8 gradient grid)
    (2, 64, 64, 32, 64, True),       # 32x32 gradient grid, NT=1 output
]


@pytest.mark.parametrize("B,Hi,Wi,cin,cout,shortcut", DGRAD_CASES)
def test_strided_data_gradient_classes(B, Hi, Wi, cin, cout, shortcut):
    Ho, Wo = Hi // 2, Wi // 2
    w_oihw, wb = _rand_weight(cout, cin, 3, seed=2)
    ws_oihw, wsb = _rand_weight(cout, cin, 1, seed=3)
    gf, gp = _rand_act(B, Ho, Wo, cout, seed=4)
    g2f, g2p = _rand_act(B, Ho, Wo, cout, seed=5)
    wd = wb.flip(1).permute(2, 1, 0).contiguous()            # [cin][9][cout], tap-reversed (what weight_prep builds)
    wsd = wsb.view(cout, cin).t().contiguous()               # [cin][cout]
    plan = ops.seg_dgrad_s2(B, Hi, Wi, cin, cout, shortcut=shortcut)
    wt = plan.tile_weights([wd, wsd] if shortcut else [wd])
    gx = ops.padded(B, Hi, Wi, cin, DEV)
    ops.interior(gx).fill_(123.0)                              # every interior pixel must be overwritten
    plan([gp, g2p] if shortcut else [gp], wt, gx)
    x = torch.zeros(B, cin, Hi, Wi, requires_grad=True)
    y = F.conv2d(x, w_oihw, stride=2, padding=1)
    tot = (y * gf.permute(0, 3, 1, 2)).sum()
    if shortcut:
        tot = tot + (F.conv2d(x, ws_oihw, stride=2) * g2f.permute(0, 3, 1, 2)).sum()
    tot.backward()
    _close_bf16(ops.interior(gx), x.grad.permute(0, 2, 3, 1), "strided data gradient")
    _border_zero(gx)


def test_dense_data_gradient_plus_shortcut():
    B, H, W, cin, cout = 3, 32, 32, 32, 160
    w_oihw, wb = _rand_weight(cout, cin, 3, seed=2)
    ws_oihw, wsb = _rand_weight(cout, cin, 1, seed=3)
    gf, gp = _rand_act(B, H, W, cout, seed=4)
    g2f, g2p = _rand_act(B, H, W, cout, seed=5)
    wd = wb.flip(1).permute(2, 1, 0).contiguous()
    wsd = wsb.view(cout, cin).t().contiguous()
    plan = ops.seg_dgrad3x3_plus_1x1(B, H, W, cin, cout)
    wt = plan.tile_weights([wd, wsd])
    gx = ops.padded(B, H, W, cin, DEV)
    plan([gp, g2p], wt, gx)
    x = torch.zeros(B, cin, H, W, requires_grad=True)
    tot = (F.conv2d(x, w_oihw, padding=1) * gf.permute(0, 3, 1, 2)).sum() + (F.conv2d(x, ws_oihw) * g2f.permute(0, 3, 1, 2)).sum()
    tot.backward()
    _close_bf16(ops.interior(gx), x.grad.permute(0, 2, 3, 1), "dense data gradient + shortcut")


def test_fp32_twin_matches_torch():
    """nbdt_ref_conv_seg (verification-only fp32 storage) on the same plans: 1e-5."""
    B, Hi, Wi, cin, cout = 2, 16, 16, 32, 64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Hi, Wi, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / 17.0
    xp = ops.padded(B, Hi, Wi, cin, DEV, torch.float32)
    ops.interior(xp).copy_(x.to(DEV))
    xs = ops.s2d_buffer(B, Hi, Wi, cin, DEV, torch.float32)
    for p in (0, 1):
        for q in (0, 1):
            ops.interior(xs)[..., (2 * p + q) * cin:(2 * p + q + 1) * cin] = ops.interior(xp)[:, p::2, q::2, :]
    plan = ops.seg_fwd_s2(B, Hi, Wi, cin, cout)
    out = ops.padded(B, Hi // 2, Wi // 2, cout, DEV, torch.float32)
    w_int = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)
    plan([xs], [w_int], out)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, stride=2, padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(ops.interior(out).cpu(), ref, rtol=1e-4, atol=1e-5)
    # strided data gradient, four classes
    gy = torch.randn(B, Hi // 2, Wi // 2, cout, generator=g)
    gp = ops.padded(B, Hi // 2, Wi // 2, cout, DEV, torch.float32)
    ops.interior(gp).copy_(gy.to(DEV))
    wd = w.permute(0, 2, 3, 1).reshape(cout, 9, cin).flip(1).permute(2, 1, 0).contiguous().to(DEV)
    dplan = ops.seg_dgrad_s2(B, Hi, Wi, cin, cout)
    gx = ops.padded(B, Hi, Wi, cin, DEV, torch.float32)
    dplan([gp], [wd.view(cin, 9 * cout)], gx)
    xt = torch.zeros(B, cin, Hi, Wi, requires_grad=True)
    (F.conv2d(xt, w, stride=2, padding=1) * gy.permute(0, 3, 1, 2)).sum().backward()
    torch.testing.assert_close(ops.interior(gx).cpu(), xt.grad.permute(0, 2, 3, 1), rtol=1e-4, atol=1e-5)


WGRAD_CASES = [   # B, Hi, Wi, cin, cout  (half-resolution grid Hi/2 x Wi/2; the new kernel needs >= 4096 output pixels)
    (16, 32, 32, 64, 160),     # 16x16 grid, WM = 5, two cin blocks
    (64, 16, 16, 32, 128),     # 8x8 grid, WM = 4
    (4, 64, 64, 32, 64),       # 32x32 grid: stages of 2 x 32, WM = 2
    (17, 32, 32, 32, 32),      # 16x16 grid, WM = 1, a pixel count that does not divide by the split
]


@pytest.mark.parametrize("B,Hi,Wi,cin,cout", WGRAD_CASES)
def test_strided_weight_gradient_over_space_to_depth(B, Hi, Wi, cin, cout):
    Ho, Wo = Hi // 2, Wi // 2
    xf, xp = _rand_act(B, Hi, Wi, cin, seed=1)
    gf, gp = _rand_act(B, Ho, Wo, cout, seed=2)
    xs = _s2d(xp, B, Hi, Wi, cin)
    w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    (F.conv2d(xf.permute(0, 3, 1, 2), w, stride=2, padding=1) * gf.permute(0, 3, 1, 2)).sum().backward()
    ref = w.grad.permute(0, 2, 3, 1).reshape(cout, 9, cin)
    d = ops.conv_wgrad_desc_s2d(B, Hi, Wi, cin, cout, 3)
    dw = torch.zeros(cout, 9, cin, device=DEV)
    ops.conv_wgrad(d, xs, gp, dw)
    assert ops.last_wgrad_kernel() == "conv_wgrad_s2d_kernel"
    tol = dict(rtol=2e-3, atol=2e-3 * ref.abs().mean().item())
    torch.testing.assert_close(dw.cpu(), ref, **tol)
    ops.conv_wgrad(d, xs, gp, dw)                     # += semantics
    torch.testing.assert_close(dw.cpu(), 2 * ref, rtol=2e-3, atol=4e-3 * ref.abs().mean().item())
    # the first-generation kernel on the same descriptor (variant 3), and both in deterministic mode: same sums
    d3 = ops.conv_wgrad_desc_s2d(B, Hi, Wi, cin, cout, 3)
    d3.variant = 3
    dw3 = torch.zeros(cout, 9, cin, device=DEV)
    ops.conv_wgrad(d3, xs, gp, dw3)
    assert ops.last_wgrad_kernel() == "conv_wgrad_dma_kernel"
    torch.testing.assert_close(dw3.cpu(), ref, **tol)
    ops.set_deterministic(True)
    try:
        a, b = torch.zeros_like(dw), torch.zeros_like(dw)
        ops.conv_wgrad(d, xs, gp, a)
        ops.conv_wgrad(d, xs, gp, b)
        assert torch.equal(a, b)
        torch.testing.assert_close(a.cpu(), ref, **tol)
    finally:
        ops.set_deterministic(False)
    # a CU-budgeted launch (atomics, fewer pixel splits)
    d.cu_budget = 96
    dwb = torch.zeros(cout, 9, cin, device=DEV)
    ops.conv_wgrad(d, xs, gp, dwb)
    torch.testing.assert_close(dwb.cpu(), ref, **tol)
    # the 1x1 stride-2 shortcut's weight gradient reads phase (0, 0) of the same tensor
    ws = torch.zeros(cout, cin, 1, 1, requires_grad=True)
    (F.conv2d(xf.permute(0, 3, 1, 2), ws, stride=2) * gf.permute(0, 3, 1, 2)).sum().backward()
    d1 = ops.conv_wgrad_desc_s2d(B, Hi, Wi, cin, cout, 1)
    dw1 = torch.zeros(cout, 1, cin, device=DEV)
    ops.conv_wgrad(d1, xs, gp, dw1)
    r1 = ws.grad.view(cout, 1, cin)
    torch.testing.assert_close(dw1.cpu(), r1, rtol=2e-3, atol=2e-3 * r1.abs().mean().item())


def _followed_by_nan(t):
    """The same tensor placed so that the memory right behind it is NaN: a tile that reaches past the end of a tensor (fewer
    images than a tile holds) must clamp every LDS-DMA piece -- what it loads for its dead pixels never reaches the output,
    but it reaches the accumulators, and NaN x 0 in the fused statistics is NaN."""
    n = t.numel()
    buf = torch.full((n + (1 << 20),), float("nan"), dtype=t.dtype, device=t.device)
    buf[:n] = t.reshape(-1)
    return buf[:n].view(t.shape)


@pytest.mark.parametrize("B,Hi,Wi,cin,cout,tile", [(5, 8, 8, 96, 160, 256), (3, 16, 16, 64, 64, 512), (9, 16, 16, 32, 160, 256)])
def test_ragged_tiles_never_read_past_the_tensors(B, Hi, Wi, cin, cout, tile):
    xf, xp = _rand_act(B, Hi, Wi, cin, seed=1)
    w_oihw, wb = _rand_weight(cout, cin, 3, seed=2)
    Ho, Wo = Hi // 2, Wi // 2
    try:
        plan = ops.seg_fwd_s2(B, Hi, Wi, cin, cout, tile=tile)
    except Exception:
        pytest.skip("shape does not fit the forced tile")
    xs = _followed_by_nan(_s2d(xp, B, Hi, Wi, cin))
    wt = plan.tile_weights([wb])
    out = ops.padded(B, Ho, Wo, cout, DEV)
    M = B * Ho * Wo
    scr = torch.zeros(((M + 255) // 256) * 2 * cout, device=DEV)
    plan([xs], wt, out, bn_scratch=scr)
    assert not torch.isnan(scr).any() and not torch.isnan(out.float()).any()
    o = ops.interior(out).float().cpu().reshape(-1, cout)
    torch.testing.assert_close(scr.view(-1, 2, cout).sum(0)[0].cpu(), o.sum(0), rtol=1e-3, atol=1e-2)
    # the data-gradient classes over a gradient tensor followed by NaN
    gf, gp = _rand_act(B, Ho, Wo, cout, seed=4)
    wd = wb.flip(1).permute(2, 1, 0).contiguous()
    dplan = ops.seg_dgrad_s2(B, Hi, Wi, cin, cout, tile=tile)
    dwt = dplan.tile_weights([wd])
    gx = ops.padded(B, Hi, Wi, cin, DEV)
    dplan([_followed_by_nan(gp)], dwt, gx)
    x = torch.zeros(B, cin, Hi, Wi, requires_grad=True)
    (F.conv2d(x, w_oihw, stride=2, padding=1) * gf.permute(0, 3, 1, 2)).sum().backward()
    _close_bf16(ops.interior(gx), x.grad.permute(0, 2, 3, 1), "strided data gradient (ragged)")
