"""Micro-benchmark of the MFMA kernels at the WRN-28-10 B=512 shapes (HIP-event timing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch
from nbdt import ops
DEV = 'cuda:0'
B = int(os.environ.get('B', 512))
which = os.environ.get('WHICH', 'wgrad,fwd,dgrad').split(',')
reps = int(os.environ.get('REPS', 5))
force = int(os.environ.get('FORCE', 0))   # desc.wide_tile: 2 = 512-pixel ping-pong kernel, 3 = 256-pixel kernels
shapes = [(32, 160, 160, 3, 1), (16, 320, 320, 3, 1), (8, 640, 640, 3, 1), (32, 160, 320, 3, 2), (32, 32, 160, 3, 1)]
if os.environ.get('SHAPES'):
    shapes = [shapes[int(i)] for i in os.environ['SHAPES'].split(',')]

def timeit(fn):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3

for (H, cin, cout, k, st) in shapes:
    Ho = H // st
    x = ops.padded(B, H, H, cin, DEV); ops.interior(x).normal_()
    g = ops.padded(B, Ho, Ho, cout, DEV); ops.interior(g).normal_()
    w = (torch.randn(cout, k * k, cin, device=DEV) * 0.05)
    wb = w.to(torch.bfloat16); wd = torch.empty(cin, k * k, cout, dtype=torch.bfloat16, device=DEV)
    ops.weight_prep(w, cout, k * k, cin, None, wd)
    out = ops.padded(B, Ho, Ho, cout, DEV); gx = ops.padded(B, H, H, cin, DEV)
    wt = ops.weight_tiles(wb) if (k == 3 and st == 1) else None
    wdt = ops.weight_tiles(wd) if (k == 3 and st == 1) else None
    dw = torch.zeros(cout, k * k, cin, device=DEV)
    flops = 2.0 * B * Ho * Ho * cout * cin * k * k
    line = f"H={H} {cin}->{cout} k{k} s{st}: "
    if 'fwd' in which:
        d = ops.conv_fwd_desc(B, H, H, cin, cout, k, st)
        if wt is not None: d.w_tiled = wt.data_ptr()
        d.wide_tile = force
        t = timeit(lambda: ops.conv_igemm(d, x, wb, out)); line += f"fwd {t*1e6:.0f}us {flops/t/1e12:.0f}TF  "
    if 'epi' in which:
        d = ops.conv_fwd_desc(B, H, H, cin, cout, k, st)
        if wt is not None: d.w_tiled = wt.data_ptr()
        d.wide_tile = force
        scr = torch.zeros(((B * Ho * Ho + 255) // 256) * 2 * cout, device=DEV)
        res = ops.padded(B, Ho, Ho, cout, DEV); ops.interior(res).normal_()
        fns = [lambda: ops.conv_igemm(d, x, wb, out), lambda: ops.conv_igemm(d, x, wb, out, residual=res),
               lambda: ops.conv_igemm(d, x, wb, out, bn_scratch=scr),
               lambda: ops.conv_igemm(d, x, wb, out, residual=res, bn_scratch=scr)]
        ts = [[], [], [], []]
        for rnd in range(3):
            for i in (2, 0, 3, 1):
                ts[i].append(timeit(fns[i]))
        line += "  ".join(f"{n} {min(t)*1e6:.0f}/{max(t)*1e6:.0f}us" for n, t in zip(("plain", "+res", "+stats", "+both"), ts))
    if 'dgrad' in which:
        ds = ops.conv_dgrad_descs(B, H, H, cin, cout, k, st)
        if wdt is not None: ds[0].w_tiled = wdt.data_ptr()
        ds[0].wide_tile = force
        t = timeit(lambda: [ops.conv_igemm(d, g, wd, gx) for d in ds]); line += f"dgrad {t*1e6:.0f}us {flops/t/1e12:.0f}TF  "
    if 'wgrad' in which:
        d = ops.conv_wgrad_desc(B, H, H, cin, cout, k, st)
        if k == 3 and st == 1: d.variant = int(os.environ.get('VARIANT', 0))
        t = timeit(lambda: ops.conv_wgrad(d, x, g, dw)); line += f"wgrad {t*1e6:.0f}us {flops/t/1e12:.0f}TF"
    print(line, flush=True)
