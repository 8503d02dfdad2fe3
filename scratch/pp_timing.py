"""Per-segment cycle sums of the ping-pong igemm kernel (NBDT_PP_TIMING build): run with NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=scratch/variants/libnbdt_tim.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import numpy as np, torch
from nbdt import ops, _C
DEV = 'cuda:0'
for (B, H, C) in [(512, 32, 160), (512, 8, 640), (8, 32, 160)]:
    x = ops.padded(B, H, H, C, DEV); ops.interior(x).normal_()
    w = (torch.randn(C, 9, C, device=DEV) * 0.05).to(torch.bfloat16)
    out = ops.padded(B, H, H, C, DEV)
    d = ops.conv_fwd_desc(B, H, H, C, C, 3, 1); d.wide_tile = 2
    wt = ops.weight_tiles(w); d.w_tiled = wt.data_ptr()
    stats = torch.zeros(((B * H * H + 255) // 256) * 2 * C, device=DEV) if os.environ.get("STATS") else None
    res = None
    if os.environ.get("RES"):
        res = ops.padded(B, H, H, C, DEV); ops.interior(res).normal_()
    for _ in range(3): ops.conv_igemm(d, x, w, out, residual=res, bn_scratch=stats)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, dtype=np.uint32)
    lib = _C.lib()
    lib.nbdt_debug_pp_timing.restype = ctypes.c_int
    rc = lib.nbdt_debug_pp_timing(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(-1, 8)
    nblk = (B * H * H // 512) * (C // 160)
    t = t[:min(nblk, 1024) * 8].astype(np.float64)
    steps = t[:, 5].mean()
    names = ["load-seg", "barrier1", "mfma-seg", "barrier2", "total", "steps", "epilogue"]
    for g, sel in (("group0", np.arange(len(t)) % 8 < 4), ("group1", np.arange(len(t)) % 8 >= 4)):
        print(f"B={B} H={H} C={C} stats={stats is not None} res={res is not None} {g}: " + "  ".join(f"{n} {t[sel, i].mean() / (steps if i < 5 else 1):7.1f}" for i, n in enumerate(names)) + f"  (cycles/step, {int(steps)} steps, rc {rc})")
    e = np.zeros(8192 * 8, dtype=np.uint32)
    lib.nbdt_debug_pp_epi.restype = ctypes.c_int
    lib.nbdt_debug_pp_epi(e.ctypes.data_as(ctypes.c_void_p))
    e = e.reshape(-1, 8)[:min(nblk, 1024) * 8].astype(np.float64)
    en = ["entry barrier+row table", "tm0 acc->LDS", "tm0 row walk", "tm1 acc->LDS", "tm1 row walk", "store acks"]
    for g, sel in (("group0", np.arange(len(e)) % 8 < 4), ("group1", np.arange(len(e)) % 8 >= 4)):
        print(f"   epilogue {g}: " + "  ".join(f"{n} {e[sel, i].mean():7.0f}" for i, n in enumerate(en)))
