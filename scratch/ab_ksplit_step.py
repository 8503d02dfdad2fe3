#!/usr/bin/env python3
"""ResNet18 training steps (configs 1 and 4) with split K of the half-tile conv kernel on / off per direction."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E, ops
from nbdt.loss import SoftTreeSupLoss
dev = torch.device("cuda", 0)
CASES = {"c1": ("CIFAR10", "induced-ResNet18", 10, 32, 1.0), "c4": ("TinyImagenet200", "induced-ResNet18", 200, 64, 10.0)}
for case in sys.argv[1:] or ["c1", "c4"]:
    dataset, hier, C, size, tsw = CASES[case]
    crit = SoftTreeSupLoss(dataset=dataset, criterion=nn.CrossEntropyLoss(), hierarchy=hier, tree_supervision_weight=tsw)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(128, 3, size, size, generator=g).to(dev)
    y = torch.randint(0, C, (128,), generator=g).to(dev)
    engines = {}
    for name, (f, d) in {"fwd+dgrad": (0, 0), "fwd only": (0, 1), "never": (1, 1)}.items():
        ops.CONV_KSPLIT.update(fwd=f, dgrad=d)
        eng = E.ResNetEngine(C, device=dev)
        for _ in range(5):
            E.train_step(eng, crit, x, y, 0.01)
        engines[name] = eng
    ops.CONV_KSPLIT.update(fwd=0, dgrad=0)
    for r in range(3):
        for name, eng in engines.items():
            for _ in range(3):
                E.train_step(eng, crit, x, y, 0.01)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                E.train_step(eng, crit, x, y, 0.01)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 40
            print(f"{case} round {r}  split K {name:10s} {ms:7.3f} ms/step  {128 / ms * 1e3:8.0f} img/s", flush=True)
