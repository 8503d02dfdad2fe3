import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss
DEV = "cuda:0"
def run(name, eng, ds, h, B, size, C):
    crit = SoftTreeSupLoss(dataset=ds, criterion=nn.CrossEntropyLoss(), hierarchy=h)
    x = torch.randn(B, 3, size, size, device=DEV); y = torch.randint(0, C, (B,), device=DEV)
    for _ in range(3): E.train_step(eng, crit, x, y, 0.01)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): E.train_step(eng, crit, x, y, 0.01)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20
    st = E.GraphedStep(eng, crit, x, y, 0.01)
    for _ in range(3): st(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): st(x, y)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20
    print(f"{name}: eager {te*1e3:.3f} ms ({B/te:.0f} img/s)  hipGraph {tg*1e3:.3f} ms ({B/tg:.0f} img/s)")
run("ResNet18 B=128 32x32", E.ResNetEngine(10, device=DEV), "CIFAR10", "induced-ResNet18", 128, 32, 10)
run("ResNet18-200 B=128 64x64", E.ResNetEngine(200, device=DEV), "TinyImagenet200", "induced-ResNet18", 128, 64, 200)
run("WRN-28-10 B=256 C=100", E.WRNEngine(100, device=DEV), "CIFAR100", "induced-wrn28_10_cifar100", 256, 32, 100)
run("WRN-28-10 B=512 C=10", E.WRNEngine(10, device=DEV), "CIFAR10", "induced-wrn28_10_cifar10", 512, 32, 10)
