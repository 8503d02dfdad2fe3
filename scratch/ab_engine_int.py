#!/usr/bin/env python3
"""Same-box A/B of one integer engine attribute on the bench workload: ab_engine_int.py <attr> <v0> <v1> [...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss
attr, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda", 0)
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(0)
x = torch.randn(512, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (512,), generator=g).to(dev)
eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
eng.set_cu_share(47.0, calibrate=False)
for _ in range(5):
    E.train_step(eng, crit, x, y, 0.01)
for r in range(3):
    for v in vals:
        eng.join_side_stream()
        setattr(eng, attr, v)
        for _ in range(4):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 30
        print(f"round {r}  {attr}={v:3d}  {ms:7.3f} ms/step  {512 / ms * 1e3:8.0f} img/s", flush=True)
