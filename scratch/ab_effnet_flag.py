#!/usr/bin/env python3
"""Same-box A/B of an EfficientNetEngine attribute at config 5 (B=128, 224x224).  usage: ab_effnet_flag.py <attr>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt.engine import train_step
from nbdt.engine_effnet import EfficientNetEngine
from nbdt.loss import SoftTreeSupLoss
attr = sys.argv[1]
dev = "cuda:0"
crit = SoftTreeSupLoss(dataset="Imagenet1000", criterion=nn.CrossEntropyLoss(), hierarchy="induced-efficientnet_b7b")
g = torch.Generator().manual_seed(0)
x = torch.randn(128, 3, 224, 224, generator=g).to(dev)
y = torch.randint(0, 1000, (128,), generator=g).to(dev)
eng = EfficientNetEngine(1000, device=dev)
for _ in range(4):
    train_step(eng, crit, x, y, 0.01)
for r in range(3):
    for val in (True, False):
        setattr(eng, attr, val)
        for _ in range(2):
            train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 20
        print(f"round {r}  {attr}={val!s:5s}  {ms:7.3f} ms/step  {128 / ms * 1e3:7.0f} img/s", flush=True)
