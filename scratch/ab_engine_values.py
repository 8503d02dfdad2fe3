#!/usr/bin/env python3
"""Same-box A/B of several values of one engine attribute on the bench workload (WRN-28-10, 512 images).
usage: ab_engine_values.py <attribute> <v1> <v2> ... [--steps 30] [--rounds 2]"""
import argparse, ast, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss
ap = argparse.ArgumentParser()
ap.add_argument("attr")
ap.add_argument("values", nargs="+")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()
dev = torch.device("cuda", 0)
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(0)
x = torch.randn(args.batch, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (args.batch,), generator=g).to(dev)
eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
eng.set_cu_share(47.0, calibrate=False)
for _ in range(5):
    E.train_step(eng, crit, x, y, 0.01)
for r in range(args.rounds):
    for v in args.values:
        setattr(eng, args.attr, ast.literal_eval(v))
        for _ in range(3):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        print(f"round {r}  {args.attr}={v:6s}  {ms:7.3f} ms/step  {args.batch / ms * 1e3:8.0f} img/s", flush=True)
