#!/usr/bin/env python3
"""Is a configuration host-bound?  Host time to ENQUEUE a training step (no synchronisation inside the loop) against the step's
wall time.  usage: host_bound_check.py [c1|c4|c3|c5|wrn]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt.engine import ResNetEngine, WRNEngine, train_step
from nbdt.engine_effnet import EfficientNetEngine
from nbdt.loss import SoftTreeSupLoss
DEV = "cuda:0"
CASES = {"c1": (lambda: ResNetEngine(10, device=DEV), "CIFAR10", "induced-ResNet18", 128, 32, 10),
         "c4": (lambda: ResNetEngine(200, device=DEV), "TinyImagenet200", "induced-ResNet18", 128, 64, 200),
         "c3": (lambda: WRNEngine(100, device=DEV), "CIFAR100", "induced-wrn28_10_cifar100", 256, 32, 100),
         "wrn": (lambda: WRNEngine(10, device=DEV), "CIFAR10", "induced-wrn28_10_cifar10", 512, 32, 10),
         "c5": (lambda: EfficientNetEngine(1000, device=DEV), "Imagenet1000", "induced-efficientnet_b7b", 128, 224, 1000)}
for name in (sys.argv[1:] or ["c1", "c4", "c5", "c3", "wrn"]):
    mk, ds, h, B, size, C = CASES[name]
    eng = mk()
    crit = SoftTreeSupLoss(dataset=ds, criterion=nn.CrossEntropyLoss(), hierarchy=h)
    x = torch.randn(B, 3, size, size, device=DEV); y = torch.randint(0, C, (B,), device=DEV)
    for _ in range(5): train_step(eng, crit, x, y, 0.01)
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N): train_step(eng, crit, x, y, 0.01)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, wall {1e3 * (t2 - t0) / N:.3f} ms/step")
    del eng
