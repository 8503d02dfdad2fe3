#!/usr/bin/env python3
"""Dense 3x3 conv on grids too small for 512-pixel tiles: 4-wave 256-pixel kernel (wide_tile 3) against the ping-pong
kernel on 256-pixel HALF tiles (5) and on 512-pixel tiles (2).  Forward + statistics, forward + residual + statistics,
data gradient + BatchNorm-backward sums."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch
from nbdt import ops
DEV = "cuda:0"
SHAPES = [(256, 8, 640), (256, 16, 320), (256, 32, 160), (128, 16, 256), (128, 8, 512), (128, 32, 128), (128, 16, 128), (128, 8, 256),
          (512, 8, 640)]
MODES = [(3, 0), (5, 0), (2, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "ksplit":
    MODES = [(5, 1), (5, 2), (5, 4), (5, 8), (0, 0)]
    sys.argv.pop(1)
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]

def timed(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for (B, H, C) in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = ops.padded(B, H, H, C, DEV); ops.interior(x).normal_()
    r = ops.padded(B, H, H, C, DEV); ops.interior(r).normal_()
    gy = ops.padded(B, H, H, C, DEV); ops.interior(gy).normal_()
    w = (torch.randn(C, 9, C, generator=g) * 0.05).to(DEV)
    wb = w.to(torch.bfloat16)
    wd = torch.empty(C, 9, C, dtype=torch.bfloat16, device=DEV)
    ops.weight_prep(w, C, 9, C, None, wd)
    wt, wdt = ops.weight_tiles(wb), ops.weight_tiles(wd)
    n_part = ((B * H * H + 255) // 256) * 2 * C
    part = torch.empty(n_part, device=DEV)
    mean, rstd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    out = ops.padded(B, H, H, C, DEV)
    fl = 2.0 * B * H * H * C * C * 9
    line = f"B={B} {H}x{H} C={C} ({fl * 1e-9:6.1f} GF):"
    for mode, ks in MODES:
        d = ops.conv_fwd_desc(B, H, H, C, C, 3, 1); d.wide_tile = mode; d.w_tiled = wt.data_ptr(); d.ksplit = ks
        (dd,) = ops.conv_dgrad_descs(B, H, H, C, C, 3, 1); dd.wide_tile = mode; dd.w_tiled = wdt.data_ptr(); dd.ksplit = ks
        try:
            t1 = timed(lambda: ops.conv_igemm(d, x, wb, out, bn_scratch=part))
            k = ops.last_igemm_kernel()
            t2 = timed(lambda: ops.conv_igemm(d, x, wb, out, residual=r, bn_scratch=part))
            t3 = timed(lambda: ops.conv_igemm_bnbwd(dd, gy, wd, out, x, mean, rstd, gamma, beta, part))
            line += f"\n   wide_tile={mode} ksplit={ks} {k:30s} fwd+stats {t1:6.1f} us ({fl / t1 * 1e-6:5.0f} TF/s)  +residual {t2:6.1f}  dgrad+bn {t3:6.1f}"
        except Exception as e:
            line += f"\n   wide_tile={mode}: {str(e)[:80]}"
    print(line, flush=True)
