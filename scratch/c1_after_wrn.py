#!/usr/bin/env python3
"""Config 1 (ResNet18 / CIFAR10, 128 images) timed the way bench.py's other_configs does it, repeatedly, inside a process that has
run WRN-28-10 steps first: looks for the slow mode (3.5 instead of 2.3 ms) seen in some bench runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss
DEV = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "wrn-first"
def timeit(fn, warm=3, steps=40):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps
if mode != "alone":
    crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
    w = E.WRNEngine(num_classes=10, device=DEV, seed=0)
    x = torch.randn(512, 3, 32, 32, device=DEV); y = torch.randint(0, 10, (512,), device=DEV)
    print("wrn", round(timeit(lambda: E.train_step(w, crit, x, y, 0.01), 5, 20), 3))
crit1 = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-ResNet18")
x1 = torch.randn(128, 3, 32, 32, device=DEV); y1 = torch.randint(0, 10, (128,), device=DEV)
for k in range(8):
    eng = E.ResNetEngine(10, device=DEV)
    if mode == "noshare":
        eng.res_share = None
    t = timeit(lambda: E.train_step(eng, crit1, x1, y1, 0.01))
    t2 = timeit(lambda: E.train_step(eng, crit1, x1, y1, 0.01), 0, 40)
    print(k, mode, round(t, 3), round(t2, 3), flush=True)
    del eng
