#!/usr/bin/env python3
"""In-step A/B of the fused head on the bench workload: three launches vs nbdt_head_soft_tree_loss by samples per block."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt._C import lib
from nbdt.loss import SoftTreeSupLoss
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
    for fused, spb in ((False, 0), (True, 16), (True, 8), (True, 4)):
        if fused:
            pass  # (the samples-per-block switch was a round-3 experiment build)
        for _ in range(3):
            E.train_step(eng, crit, x, y, 0.01, fused_head=fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            E.train_step(eng, crit, x, y, 0.01, fused_head=fused)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 40
        print(f"round {r}  fused_head={fused!s:5s} samples/block<={spb:2d}  {ms:7.3f} ms/step", flush=True)
