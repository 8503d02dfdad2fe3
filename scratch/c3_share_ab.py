"""C3 (WRN-28-10 / CIFAR100, 256 images): CU sharing off / forced / calibrated, alternating."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt.engine import WRNEngine, train_step
from nbdt.loss import SoftTreeSupLoss
DEV = "cuda:0"
B = int(os.environ.get("B", 256))
eng = WRNEngine(100, device=DEV)
crit = SoftTreeSupLoss(dataset="CIFAR100", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar100")
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, 32, 32, generator=g).to(DEV); y = torch.randint(0, 100, (B,), generator=g).to(DEV)
def run(tag, setup):
    setup()
    for _ in range(4): train_step(eng, crit, x, y, 0.01)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): train_step(eng, crit, x, y, 0.01)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{tag:12s} {dt*1e3:7.3f} ms  {B/dt:8.0f} img/s  {eng.cu_share_report}", flush=True)
for r in range(2):
    run("off", lambda: eng.set_cu_share(None))
    run("forced", lambda: eng.set_cu_share(47.0, calibrate=False))
    run("calibrated", lambda: eng.set_cu_share(47.0))
