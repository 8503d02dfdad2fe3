#!/usr/bin/env python3
"""cProfile of the HOST side of ResNet18 / CIFAR10 training steps (config 1, which is bound by it)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt.engine import ResNetEngine, train_step
from nbdt.loss import SoftTreeSupLoss
DEV = "cuda:0"
eng = ResNetEngine(10, device=DEV)
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-ResNet18")
x = torch.randn(128, 3, 32, 32, device=DEV); y = torch.randint(0, 10, (128,), device=DEV)
for _ in range(5): train_step(eng, crit, x, y, 0.01)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50): train_step(eng, crit, x, y, 0.01)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
