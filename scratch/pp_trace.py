"""Timeline of the ping-pong igemm kernel per CU (NBDT_PP_TIMING=2 build): run with NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=scratch/variants/libnbdt_trace.so.
Four s_memtime stamps per block (entry, K loop start, epilogue start, stores acknowledged) + HW_ID/XCC_ID; blocks are
grouped by CU and sorted by entry time: the gap between one block's end and the next block's entry is dispatch cost."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import numpy as np, torch
from nbdt import ops, _C
DEV = 'cuda:0'
for (B, H, C) in [(512, 32, 160), (512, 16, 320), (512, 8, 640)]:
    x = ops.padded(B, H, H, C, DEV); ops.interior(x).normal_()
    w = (torch.randn(C, 9, C, device=DEV) * 0.05).to(torch.bfloat16)
    out = ops.padded(B, H, H, C, DEV)
    d = ops.conv_fwd_desc(B, H, H, C, C, 3, 1); d.wide_tile = 2
    wt = ops.weight_tiles(w); d.w_tiled = wt.data_ptr()
    for _ in range(3): ops.conv_igemm(d, x, w, out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.conv_igemm(d, x, w, out); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3
    buf = np.zeros(8192 * 8, dtype=np.uint32)
    lib = _C.lib(); lib.nbdt_debug_pp_timing.restype = ctypes.c_int
    lib.nbdt_debug_pp_timing(buf.ctypes.data_as(ctypes.c_void_p))
    nblk = (B * H * H // 512) * (C // 160)
    t = buf.reshape(-1, 8)[:nblk].astype(np.int64)
    hw, xcc = t[:, 0], t[:, 1] & 0xf
    cu = (xcc << 16) | (hw & 0xff00)          # HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    ent, loop, epi, end = t[:, 2], t[:, 3], t[:, 4], t[:, 5]
    print(f"== B={B} H={H} C={C}: {nblk} tiles on {len(set(cu))} CUs, kernel {us:.1f} us")
    d32 = lambda a, b: (a - b) & 0xffffffff
    print(f"   per tile: prologue {np.mean(d32(loop, ent)):.0f}  K loop {np.mean(d32(epi, loop)):.0f}  "
          f"epilogue+acks {np.mean(d32(end, epi)):.0f}  total {np.mean(d32(end, ent)):.0f}")
