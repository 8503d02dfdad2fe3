"""ResNetEngine with the strided blocks on the slice-list kernels vs rounds 1-5's launches: same weights, same batch,
deterministic mode -- losses / gradients to bf16 noise; then C1 / C4 step times both ways."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine, ops
from nbdt.loss import SoftTreeSupLoss
g = torch.Generator().manual_seed(0)
for (name, classes, size, ds, hier, tsw) in [("C1", 10, 32, "CIFAR10", "induced-ResNet18", 1.0), ("C4", 200, 64, "TinyImagenet200", "induced-ResNet18", 10.0)]:
    crit = SoftTreeSupLoss(dataset=ds, criterion=nn.CrossEntropyLoss(), hierarchy=hier, tree_supervision_weight=tsw)
    img = torch.randn(128, 3, size, size, generator=g).cuda(); y = torch.randint(0, classes, (128,), generator=g).cuda()
    res = {}
    ops.set_deterministic(True)
    for seg in (False, True):
        eng = engine.ResNetEngine(num_classes=classes, device="cuda:0", seed=0)
        eng.use_seg = seg
        loss = engine.train_step(eng, crit, img, y, lr=0.0, zero_grad=False)
        torch.cuda.synchronize()
        res[seg] = (loss.item(), eng.store.grad.clone(), [b.get("seg") for b in eng.blocks])
    ops.set_deterministic(False)
    (l0, g0, _), (l1, g1, segs) = res[False], res[True]
    cos = torch.nn.functional.cosine_similarity(g0, g1, dim=0).item()
    print(f"{name}: loss {l0:.6f} vs {l1:.6f}; grad cos {cos:.5f}, rel L2 {((g0 - g1).norm() / g0.norm()).item():.4f}; seg blocks {segs}")
    for seg in (False, True, False, True):
        eng = engine.ResNetEngine(num_classes=classes, device="cuda:0", seed=0)
        eng.use_seg = seg
        for _ in range(5): engine.train_step(eng, crit, img, y, lr=0.01)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): engine.train_step(eng, crit, img, y, lr=0.01)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
        print(f"  {name} use_seg={seg}: {dt*1e3:.3f} ms/step, {128/dt:.0f} img/s", flush=True)
        del eng
