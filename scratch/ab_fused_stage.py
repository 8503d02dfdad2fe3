#!/usr/bin/env python3
"""Same-box A/B: which stages take their BatchNorm-backward sums from the data gradient's epilogue (engine.share_fused_hw)
and how many CUs the confined elementwise pass then gets (engine.share_fused_target_us).  WRN-28-10, 512 images.
usage: ab_fused_stage.py [--steps 30] [--rounds 2]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss
ap = argparse.ArgumentParser()
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
configs = [(0, 200.0), (1024, 200.0), (1024, 150.0), (1024, 260.0), (1024, 320.0), (256, 200.0), (64, 200.0)]
for r in range(args.rounds):
    for hw, us in configs:
        eng.share_fused_hw, eng.share_fused_target_us = hw, us
        for _ in range(3):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        print(f"round {r}  share_fused_hw={hw:5d} target_us={us:5.0f}  {ms:7.3f} ms/step  {args.batch / ms * 1e3:8.0f} img/s", flush=True)
