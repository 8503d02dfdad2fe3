#!/usr/bin/env python3
"""Same-box A/B of one EfficientNetEngine attribute on BASELINE config 5 (EfficientNet-B0 + SoftTreeSupLoss, 128 x 224 x 224).
usage: ab_effnet_flag.py <attribute> [--steps 15] [--rounds 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt.engine import train_step
from nbdt.engine_effnet import EfficientNetEngine
from nbdt.loss import SoftTreeSupLoss
ap = argparse.ArgumentParser()
ap.add_argument("attr")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=15)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
eng = EfficientNetEngine(num_classes=1000, device="cuda:0")
crit = SoftTreeSupLoss(dataset="Imagenet1000", criterion=nn.CrossEntropyLoss(), hierarchy="induced-efficientnet_b7b")
g = torch.Generator().manual_seed(0)
x = torch.randn(a.batch, 3, 224, 224, generator=g).cuda()
y = torch.randint(0, 1000, (a.batch,), generator=g).cuda()
for _ in range(3):
    train_step(eng, crit, x, y, lr=0.01)
for r in range(a.rounds):
    for val in (True, False):
        setattr(eng, a.attr, val)
        for _ in range(2):
            train_step(eng, crit, x, y, lr=0.01)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = train_step(eng, crit, x, y, lr=0.01)
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / a.steps
        print(f"round {r}  {a.attr}={val!s:5s}  {ms:7.3f} ms/step  {a.batch / ms * 1e3:8.0f} img/s  loss {loss.item():.4f}", flush=True)
