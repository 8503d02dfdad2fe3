import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss
eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device="cuda:0", seed=0)
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(0)
img = torch.randn(512, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 10, (512,), generator=g).cuda()
rows = []
for i in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    E.train_step(eng, crit, img, y, 0.01)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
print("per-step (cpu enqueue ms, wall ms):", " ".join(f"({a:.1f},{b:.1f})" for a, b in rows))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    E.train_step(eng, crit, img, y, 0.01)
torch.cuda.synchronize(); print("10 back-to-back steps: %.2f ms/step" % (1e2 * (time.perf_counter() - t0)))
