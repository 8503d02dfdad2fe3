"""Sanity: memorise one batch for many steps (races / stale buffers would show up as divergence or NaN)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.engine_effnet import EfficientNetEngine
from nbdt.loss import SoftTreeSupLoss, HardTreeSupLoss
DEV = "cuda:0"
def run(name, eng, crit, B, size, C, steps, lr):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, size, size, generator=g).to(DEV); y = torch.randint(0, C, (B,), generator=g).to(DEV)
    ls = []
    for i in range(steps):
        ls.append(E.train_step(eng, crit, x, y, lr))
    ls = [l.item() for l in ls]
    z = eng.forward(x, training=False)
    acc = (z.argmax(1) == y).float().mean().item()
    print(f"{name}: loss {ls[0]:.3f} -> {ls[len(ls)//2]:.3f} -> {ls[-1]:.4f}; finite={all(l == l for l in ls)}; eval-mode train-batch acc {acc:.3f}")
ce = nn.CrossEntropyLoss()
run("WRN-28-10 B=128", E.WRNEngine(10, device=DEV), SoftTreeSupLoss(dataset="CIFAR10", criterion=ce, hierarchy="induced-wrn28_10_cifar10"), 128, 32, 10, 80, 0.02)
run("ResNet18 hard loss B=128", E.ResNetEngine(10, device=DEV), HardTreeSupLoss(dataset="CIFAR10", criterion=ce, hierarchy="induced-ResNet18"), 128, 32, 10, 80, 0.02)
run("EfficientNet-B0 B=32 96px", EfficientNetEngine(1000, dropout_rate=0.2, device=DEV), SoftTreeSupLoss(dataset="Imagenet1000", criterion=ce, hierarchy="induced-efficientnet_b7b"), 32, 96, 1000, 80, 0.02)
