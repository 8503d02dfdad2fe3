#!/usr/bin/env python3
"""Headline step with the weight-gradient stream at another HIP priority, and with the main work on a high-priority stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss
dev = torch.device("cuda", 0)
print("priority range", torch.cuda.Stream.priority_range(), flush=True)
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(0)
x = torch.randn(512, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (512,), generator=g).to(dev)
engines = {}
lo, hi = torch.cuda.Stream.priority_range()
for name, (side, main) in {"side normal": (0, None), "side low": (lo, None), "side high": (hi, None), "main high": (0, hi)}.items():
    E.SIDE_STREAM_PRIORITY = side
    eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
    eng.set_cu_share(47.0, calibrate=False)
    ms_ = torch.cuda.Stream(device=dev, priority=main) if main is not None else torch.cuda.current_stream(dev)
    with torch.cuda.stream(ms_):
        for _ in range(4):
            E.train_step(eng, crit, x, y, 0.01)
    torch.cuda.synchronize()
    engines[name] = (eng, ms_)
for r in range(3):
    for name, (eng, ms_) in engines.items():
        with torch.cuda.stream(ms_):
            for _ in range(3):
                E.train_step(eng, crit, x, y, 0.01)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(25):
                E.train_step(eng, crit, x, y, 0.01)
            torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 25
        print(f"round {r}  {name:12s} {ms:7.3f} ms/step  {512 / ms * 1e3:8.0f} img/s", flush=True)
