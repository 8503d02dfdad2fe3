"""Per-segment cycle sums of conv_seg_kernel (NBDT_SEG_TIMING build): NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=scratch/variants/libnbdt_segtim.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import numpy as np, torch
from nbdt import ops, _C
DEV = 'cuda:0'
B = int(os.environ.get('B', 512))
TILE = int(os.environ.get('TILE', 0)); NBUF = int(os.environ.get('NBUF', 0))

def act(H, C): t = ops.padded(B, H, H, C, DEV); ops.interior(t).normal_(); return t

def dump(name, plan, run):
    for _ in range(3): run()
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, dtype=np.uint32)
    lib = _C.lib()
    lib.nbdt_debug_seg_timing.restype = ctypes.c_int
    lib.nbdt_debug_seg_timing(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(-1, 8)[:1024 * 8].astype(np.float64)
    t = t[t[:, 5] > 0]
    for ns in sorted(set(t[:, 5].astype(int))):
        u = t[t[:, 5] == ns]
        w = np.arange(len(u))
        line = f"{name} steps {ns:3d} ({len(u)//8} items): "
        names = ["L", "b1", "M", "b2", "loop"]
        line += "  ".join(f"{n} {u[:, i].mean() / ns:6.1f}" for i, n in enumerate(names))
        line += f"  | prologue {u[:, 6].mean():6.0f}  epilogue {u[:, 7].mean():6.0f}  loop total {u[:, 4].mean():7.0f}"
        print(line, flush=True)

for (Hi, cin, cout) in [(32, 160, 320), (16, 320, 640)]:
    Ho = Hi // 2
    xs = ops.s2d_buffer(B, Hi, Hi, cin, DEV); ops.interior(xs).normal_()
    wb = (torch.randn(cout, 9, cin, device=DEV) * 0.05).to(torch.bfloat16)
    wd = wb.flip(1).permute(2, 1, 0).contiguous()
    wsb = (torch.randn(cout, cin, device=DEV) * 0.05).to(torch.bfloat16)
    M = B * Ho * Ho
    scr = torch.zeros(((M + 255) // 256) * 2 * cout, device=DEV)
    out = ops.padded(B, Ho, Ho, cout, DEV)
    plan = ops.seg_fwd_s2(B, Hi, Hi, cin, cout, tile=TILE, nbuf=NBUF)
    wt = plan.tile_weights([wb])
    dump(f"fwd s2 {Hi}", plan, lambda: plan([xs], wt, out, bn_scratch=scr))
    a2 = act(Ho, cout); w2b = (torch.randn(cout, 9, cout, device=DEV) * 0.05).to(torch.bfloat16)
    plan2 = ops.seg_conv3x3_plus_1x1(B, Ho, Ho, cout, cout, cin, 4 * cin, tile=TILE, nbuf=NBUF)
    wt2 = plan2.tile_weights([w2b, wsb])
    dump(f"conv2+sc {Ho}", plan2, lambda: plan2([a2, xs], wt2, out, bn_scratch=scr))
    g = act(Ho, cout); g2 = act(Ho, cout); gx = ops.padded(B, Hi, Hi, cin, DEV)
    plan3 = ops.seg_dgrad_s2(B, Hi, Hi, cin, cout, shortcut=True, tile=TILE, nbuf=NBUF)
    wt3 = plan3.tile_weights([wd.view(cin, 9 * cout), wsb.t().contiguous()])
    dump(f"dgrad {Hi}", plan3, lambda: plan3([g, g2], wt3, gx))
