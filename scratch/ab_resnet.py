#!/usr/bin/env python3
"""ResNet18 training throughput at the two ResNet configurations (C1: CIFAR10 32x32 B=128, C4: TinyImagenet200 64x64 B=128)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss
dev = torch.device("cuda", 0)
for ds, h, C, size, B in (("CIFAR10", "induced-ResNet18", 10, 32, 128), ("TinyImagenet200", "induced-ResNet18", 200, 64, 128),
                          ("CIFAR10", "induced-ResNet18", 10, 32, 512)):
    crit = SoftTreeSupLoss(dataset=ds, criterion=nn.CrossEntropyLoss(), hierarchy=h, tree_supervision_weight=10.0 if C == 200 else 1.0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, size, size, generator=g).to(dev)
    y = torch.randint(0, C, (B,), generator=g).to(dev)
    eng = E.ResNetEngine(C, device=dev)
    for _ in range(5):
        E.train_step(eng, crit, x, y, 0.01)
    ts = []
    for r in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0) / 40)
    print(f"{ds} {size}x{size} B={B}: " + " / ".join(f"{t:.3f}" for t in ts) + f" ms/step  {B / min(ts) * 1e3:.0f} img/s", flush=True)
