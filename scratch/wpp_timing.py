"""Per-segment cycle sums of the ping-pong weight-gradient kernel (NBDT_WPP_TIMING build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import numpy as np, torch
from nbdt import ops, _C
DEV = 'cuda:0'
for (B, H, C) in [(512, 32, 160), (512, 8, 640)]:
    x = ops.padded(B, H, H, C, DEV); ops.interior(x).normal_()
    g = ops.padded(B, H, H, C, DEV); ops.interior(g).normal_()
    dw = torch.zeros(C, 9, C, device=DEV)
    d = ops.conv_wgrad_desc(B, H, H, C, C, 3, 1); d.variant = 2
    for _ in range(3): ops.conv_wgrad(d, x, g, dw)
    torch.cuda.synchronize()
    buf = np.zeros(2048 * 8, dtype=np.uint32)
    lib = _C.lib(); lib.nbdt_debug_wpp_timing.restype = ctypes.c_int
    lib.nbdt_debug_wpp_timing(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(-1, 8)[:256 * 8].astype(np.float64)
    t = t[t[:, 5] > 0]
    names = ["load-seg", "barrier1", "mfma-seg", "barrier2", "total"]
    for gname, sel in (("group0", np.arange(len(t)) % 8 < 4), ("group1", np.arange(len(t)) % 8 >= 4)):
        st = t[sel, 5].mean()
        print(f"B={B} H={H} C={C} {gname}: " + "  ".join(f"{n} {(t[sel, i] / t[sel, 5]).mean():7.1f}" for i, n in enumerate(names)) + f"  (cycles/stage, {st:.0f} stages)")
