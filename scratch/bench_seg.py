"""Slice-list conv launches (csrc/conv_seg.hip) at the WRN-28-10 / 512-image shapes of the three shape-changing units,
each beside the launches it replaces (HIP-event timing, alone on the GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch
from nbdt import ops
DEV = 'cuda:0'
B = int(os.environ.get('B', 512))
reps = int(os.environ.get('REPS', 10))
TILE = int(os.environ.get('TILE', 0)); NBUF = int(os.environ.get('NBUF', 0))
which = os.environ.get('WHICH', 'fwd,plus,dgrad,first').split(',')

def timeit(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps * 1e-3)
    return best

def act(H, C): t = ops.padded(B, H, H, C, DEV); ops.interior(t).normal_(); return t
def wts(cout, taps, cin):
    w = torch.randn(cout, taps, cin, device=DEV) * 0.05
    wb = w.to(torch.bfloat16); wd = torch.empty(cin, taps, cout, dtype=torch.bfloat16, device=DEV)
    ops.weight_prep(w, cout, taps, cin, None, wd)
    return wb, wd

def report(name, flops, t_new, t_old, plan):
    print(f"{name:28s} new {t_new*1e6:6.1f} us {flops/t_new/1e12:6.0f} TF ({flops/t_new/2.5e15:.3f})   old {t_old*1e6:6.1f} us   "
          f"tile {plan.tile} nbuf {plan.nbuf} steps {plan.nsteps} max rounds {plan.max_rounds}", flush=True)

for (Hi, cin, cout) in [(32, 160, 320), (16, 320, 640)]:
    Ho = Hi // 2
    x = act(Hi, cin); xs = ops.s2d_buffer(B, Hi, Hi, cin, DEV); ops.interior(xs).normal_()
    wb, wd = wts(cout, 9, cin); wsb, wsd = wts(cout, 1, cin)
    M = B * Ho * Ho
    scr = torch.zeros(((M + 255) // 256) * 2 * cout, device=DEV)
    out = ops.padded(B, Ho, Ho, cout, DEV)
    if 'fwd' in which:
        plan = ops.seg_fwd_s2(B, Hi, Hi, cin, cout, tile=TILE, nbuf=NBUF)
        wt = plan.tile_weights([wb])
        d = ops.conv_fwd_desc(B, Hi, Hi, cin, cout, 3, 2)
        report(f"fwd s2 {Hi}x{Hi} {cin}->{cout}", plan.flops, timeit(lambda: plan([xs], wt, out, bn_scratch=scr)),
               timeit(lambda: ops.conv_igemm(d, x, wb, out, bn_scratch=scr)), plan)
    if 'plus' in which:
        a2 = act(Ho, cout); w2b, w2d = wts(cout, 9, cout)
        plan = ops.seg_conv3x3_plus_1x1(B, Ho, Ho, cout, cout, cin, 4 * cin, tile=TILE, nbuf=NBUF)
        wt = plan.tile_weights([w2b, wsb.view(cout, cin)])
        d2 = ops.conv_fwd_desc(B, Ho, Ho, cout, cout, 3, 1); d2.w_tiled = ops.weight_tiles(w2b).data_ptr()
        d1 = ops.conv_fwd_desc(B, Hi, Hi, cin, cout, 1, 2)
        idn = ops.padded(B, Ho, Ho, cout, DEV)
        def old():
            ops.conv_igemm(d1, x, wsb, idn)
            ops.conv_igemm(d2, a2, w2b, out, residual=idn, bn_scratch=scr)
        report(f"conv2+sc {Ho}x{Ho} {cout}", plan.flops, timeit(lambda: plan([a2, xs], wt, out, bn_scratch=scr)), timeit(old), plan)
    if 'dgrad' in which:
        g = act(Ho, cout); g2 = act(Ho, cout); gx = ops.padded(B, Hi, Hi, cin, DEV)
        plan = ops.seg_dgrad_s2(B, Hi, Hi, cin, cout, shortcut=True, tile=TILE, nbuf=NBUF)
        wt = plan.tile_weights([wd.view(cin, 9 * cout), wsd.view(cin, cout)])
        ds = ops.conv_dgrad_descs(B, Hi, Hi, cin, cout, 3, 2)
        d1 = ops.conv_dgrad_descs(B, Hi, Hi, cin, cout, 1, 2, accumulate=True)[0]
        def old():
            ops.conv_igemm_multi(ds, g, wd, gx)
            ops.conv_igemm(d1, g2, wsd, gx)
        report(f"dgrad s2+sc {Hi}x{Hi} {cout}->{cin}", plan.flops, timeit(lambda: plan([g, g2], wt, gx)), timeit(old), plan)

if 'first' in which:
    H, cin, cout = 32, 32, 160
    a1 = act(H, cin); a2 = act(H, cout)
    w2b, w2d = wts(cout, 9, cout); wsb, wsd = wts(cout, 1, cin); w1b, w1d = wts(cout, 9, cin)
    out = ops.padded(B, H, H, cout, DEV); M = B * H * H
    scr = torch.zeros(((M + 255) // 256) * 2 * cout, device=DEV)
    plan = ops.seg_conv3x3_plus_1x1(B, H, H, cout, cout, cin, cin, tile=TILE, nbuf=NBUF)
    wt = plan.tile_weights([w2b, wsb.view(cout, cin)])
    d2 = ops.conv_fwd_desc(B, H, H, cout, cout, 3, 1); d2.w_tiled = ops.weight_tiles(w2b).data_ptr()
    d1 = ops.conv_fwd_desc(B, H, H, cin, cout, 1, 1)
    idn = ops.padded(B, H, H, cout, DEV)
    def old():
        ops.conv_igemm(d1, a1, wsb, idn)
        ops.conv_igemm(d2, a2, w2b, out, residual=idn, bn_scratch=scr)
    report("s1u1 conv2+sc 32x32 160", plan.flops, timeit(lambda: plan([a2, a1], wt, out, bn_scratch=scr)), timeit(old), plan)
    g = act(H, cout); g2 = act(H, cout); gx = ops.padded(B, H, H, cin, DEV)
    plan = ops.seg_dgrad3x3_plus_1x1(B, H, H, cin, cout, tile=TILE, nbuf=NBUF)
    wt = plan.tile_weights([w1d.view(cin, 9 * cout), wsd.view(cin, cout)])
    dd = ops.conv_dgrad_descs(B, H, H, cin, cout, 3, 1)[0]; dd.w_tiled = ops.weight_tiles(w1d).data_ptr()
    ds = ops.conv_dgrad_descs(B, H, H, cin, cout, 1, 1, accumulate=True)[0]
    def old():
        ops.conv_igemm(dd, g, w1d, gx)
        ops.conv_igemm(ds, g2, wsd, gx)
    report("s1u1 dgrad+sc 32x32 160->32", plan.flops, timeit(lambda: plan([g, g2], wt, gx)), timeit(old), plan)

if 'wgrad' in which:
    for (Hi, cin, cout) in [(32, 160, 320), (16, 320, 640)]:
        Ho = Hi // 2
        xs = ops.s2d_buffer(B, Hi, Hi, cin, DEV); ops.interior(xs).normal_()
        x = act(Hi, cin); g = act(Ho, cout)
        dw = torch.zeros(cout, 9, cin, device=DEV)
        dn = ops.conv_wgrad_desc_s2d(B, Hi, Hi, cin, cout, 3)
        d3 = ops.conv_wgrad_desc_s2d(B, Hi, Hi, cin, cout, 3); d3.variant = 3
        do = ops.conv_wgrad_desc(B, Hi, Hi, cin, cout, 3, 2)
        fl = 2.0 * B * Ho * Ho * 9 * cin * cout
        tn = timeit(lambda: ops.conv_wgrad(dn, xs, g, dw)); k = ops.last_wgrad_kernel()
        t3 = timeit(lambda: ops.conv_wgrad(d3, xs, g, dw))
        to = timeit(lambda: ops.conv_wgrad(do, x, g, dw))
        print(f"wgrad s2 {Hi}x{Hi} {cin}->{cout}: {k} {tn*1e6:6.1f} us {fl/tn/1e12:5.0f} TF ({fl/tn/2.5e15:.3f})   wgrad_dma on s2d {t3*1e6:6.1f} us   wgrad_dma plain {to*1e6:6.1f} us", flush=True)
