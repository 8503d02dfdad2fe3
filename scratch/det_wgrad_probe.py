import os, sys
sys.path.insert(0, os.getcwd())
import nbdt_path; nbdt_path.add()
import torch
from nbdt import ops
DEV='cuda:0'; B=512
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps*1e3
for (H,C) in [(32,160),(16,320),(8,640)]:
    x=ops.padded(B,H,H,C,DEV); ops.interior(x).normal_()
    g=ops.padded(B,H,H,C,DEV); ops.interior(g).normal_()
    dw=torch.zeros(C,9,C,device=DEV)
    d=ops.conv_wgrad_desc(B,H,H,C,C,3,1)
    t0=timeit(lambda: ops.conv_wgrad(d,x,g,dw))
    ops.set_deterministic(True)
    t1=timeit(lambda: ops.conv_wgrad(d,x,g,dw))
    ops.set_deterministic(False)
    print(f"H={H} C={C}: wgrad {t0:.0f} us; deterministic mode (memset + uncontended atomics into per-split copies + fold) {t1:.0f} us")
