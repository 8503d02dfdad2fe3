"""Quick check of the slice-list path inside WRNEngine: a step with use_seg on vs off on a small WRN (same weights, same
batch) -- losses and gradients must agree to bf16 noise; then the benched configuration's step time both ways."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine, ops
from nbdt.loss import SoftTreeSupLoss
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(0)
if os.environ.get("SMALL", "1") == "1":
    img = torch.randn(16, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 10, (16,), generator=g).cuda()
    res = {}
    for seg in (False, True):
        eng = engine.WRNEngine(num_classes=10, blocks=10, width_factor=2, device="cuda:0", seed=0)
        eng.use_seg = seg
        ops.set_deterministic(True)
        loss = engine.train_step(eng, crit, img, y, lr=0.0, zero_grad=False)
        torch.cuda.synchronize()
        res[seg] = (loss.item(), eng.store.grad.clone())
    ops.set_deterministic(False)
    (l0, g0), (l1, g1) = res[False], res[True]
    cos = torch.nn.functional.cosine_similarity(g0, g1, dim=0).item()
    print(f"loss {l0:.6f} vs {l1:.6f}; grad cos {cos:.5f}, rel L2 {((g0 - g1).norm() / g0.norm()).item():.4f}")
    for name, (off, shape) in eng.store.entries.items():
        n = 1
        for q in shape: n *= q
        a, b = g0[off:off + n], g1[off:off + n]
        if a.norm() > 0:
            c = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
            if c < 0.98: print(f"   {name}: cos {c:.4f} norms {a.norm():.4g} {b.norm():.4g}")
B = int(os.environ.get("B", 512))
img = torch.randn(B, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 10, (B,), generator=g).cuda()
cfgs = [(False, False, None), (True, True, None)]
for seg, share, join in cfgs * 6:
    eng = engine.WRNEngine(num_classes=10, blocks=28, width_factor=10, device="cuda:0", seed=0)
    eng.use_seg = seg
    eng.seg_share = share
    eng.seg_join = join
    eng.set_cu_share(47.0, calibrate=False)
    for _ in range(4): engine.train_step(eng, crit, img, y, lr=0.01)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): engine.train_step(eng, crit, img, y, lr=0.01)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"use_seg={seg} seg_share={share} seg_join={join}: {dt*1e3:.3f} ms/step, {B/dt:.0f} img/s", flush=True)
    del eng
