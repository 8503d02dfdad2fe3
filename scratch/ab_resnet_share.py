#!/usr/bin/env python3
"""Same-box A/B of ResNetEngine.res_share on BASELINE config 4 (ResNet18, TinyImagenet200 64x64) or config 1 (CIFAR10 32x32).
usage: ab_resnet_share.py [--config c4|c1] [--batch 128] [--steps 40] [--rounds 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt.engine import ResNetEngine, train_step
from nbdt.loss import SoftTreeSupLoss
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c4")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--targets", default="None,30,45,60,80")
ap.add_argument("--bn2", default="", help="comma list of res_share_bn2 values to sweep (res_share stays at its default)")
a = ap.parse_args()
DEV = "cuda:0"
if a.config == "c4":
    eng, ds, size, C, tsw = ResNetEngine(200, device=DEV), "TinyImagenet200", 64, 200, 10.0
else:
    eng, ds, size, C, tsw = ResNetEngine(10, device=DEV), "CIFAR10", 32, 10, 1.0
crit = SoftTreeSupLoss(dataset=ds, criterion=nn.CrossEntropyLoss(), hierarchy="induced-ResNet18", tree_supervision_weight=tsw)
g = torch.Generator().manual_seed(0)
x = torch.randn(a.batch, 3, size, size, generator=g).to(DEV)
y = torch.randint(0, C, (a.batch,), generator=g).to(DEV)
for _ in range(5):
    train_step(eng, crit, x, y, 0.01)
vals = [None if v == "None" else float(v) for v in a.targets.split(",")]
if a.bn2:
    vals = [float(v) for v in a.bn2.split(",")]
for r in range(a.rounds):
    for v in vals:
        if a.bn2:
            eng.res_share_bn2 = v
        else:
            eng.res_share = None if v is None else (47.0, v, 16, 128)
        for _ in range(3):
            train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / a.steps
        print(f"round {r}  {'res_share_bn2' if a.bn2 else 'res_share target'} {v!s:5s}  {ms:7.3f} ms/step  {a.batch / ms * 1e3:8.0f} img/s", flush=True)
