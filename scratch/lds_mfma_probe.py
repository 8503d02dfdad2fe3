"""LDS-operand MFMA streams (nbdt_probe_lds_mfma): production tiling vs one wave per SIMD, beside the register-only stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch
from nbdt._C import lib, check, ptr
from nbdt.ops import stream_ptr
dev = torch.device('cuda:0')
sink = torch.zeros(4, device=dev)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
for rnd in range(3):
    t = timed(lambda: check(lib().nbdt_probe_mfma_stream(256, 6000, ptr(sink), stream_ptr(dev))))
    line = f"round {rnd}: register-only stream {256*8*6000*16*32768.0/t/1e12:7.1f} TF/s"
    for v, name in ((0, "8 waves x 64 px, 14 reads / 20 MFMA"), (2, "same, 10 reads"), (3, "same, 6 reads"), (1, "4 waves x 128 px, 18 / 40")):
        iters = 4000
        t = timed(lambda: check(lib().nbdt_probe_lds_mfma(256, iters, v, ptr(sink), stream_ptr(dev))))
        line += f"   {name}: {256*iters*160*32768.0/t/1e12:7.1f} TF/s"
    print(line, flush=True)
