"""nbdt_se_gate_fwd / nbdt_se_gate_bwd at EfficientNet-B0's SE blocks (batch 128): HIP events over 50 calls."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "neural-backed-decision-trees_amd"))
import torch
from nbdt import ops
DEV = "cuda:0"
B = 128
tot = 0.0
for Cr, S in ((32, 8), (96, 4), (144, 6), (240, 10), (480, 20), (672, 28), (1152, 48)):
    C = Cr
    g = torch.Generator().manual_seed(Cr)
    pooled = torch.randn(B, C, generator=g).to(DEV)
    w1 = (torch.randn(S, Cr, generator=g) * 0.1).to(DEV); b1 = torch.randn(S, generator=g).to(DEV)
    w2 = (torch.randn(Cr, S, generator=g) * 0.1).to(DEV); b2 = torch.randn(Cr, generator=g).to(DEV)
    pre1, gate = torch.empty(B, S, device=DEV), torch.empty(B, C, device=DEV)
    dgate = torch.randn(B, C, generator=g).to(DEV)
    dpre2, dpre1, gpool = torch.empty(B, Cr, device=DEV), torch.empty(B, S, device=DEV), torch.empty(B, C, device=DEV)
    dw1, db1, dw2, db2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros_like(b2)
    fwd = lambda: ops.se_gate_fwd(pooled, w1, b1, w2, b2, pre1, gate, Cr)
    bwd = lambda: ops.se_gate_bwd(dgate, gate, pre1, pooled, w1, w2, dpre2, dpre1, gpool, dw1, db1, dw2, db2, Cr)
    res = []
    for fn in (fwd, bwd):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    tot += sum(res)
    print(f"Cr={Cr:5d} S={S:3d}: fwd {res[0]:6.1f} us   bwd (gate + parameter gradients) {res[1]:6.1f} us", flush=True)
print(f"sum {tot:.1f} us")
