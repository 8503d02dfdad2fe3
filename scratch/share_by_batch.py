#!/usr/bin/env python3
"""What the CU-sharing calibration decides, and what each schedule costs, by batch size and class count (WRN-28-10)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss
dev = torch.device("cuda", 0)
for B, C, ds, h in ((128, 10, "CIFAR10", "induced-wrn28_10_cifar10"), (256, 100, "CIFAR100", "induced-wrn28_10_cifar100"),
                    (384, 10, "CIFAR10", "induced-wrn28_10_cifar10"), (512, 10, "CIFAR10", "induced-wrn28_10_cifar10")):
    crit = SoftTreeSupLoss(dataset=ds, criterion=nn.CrossEntropyLoss(), hierarchy=h)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, C, (B,), generator=g).to(dev)
    eng = E.WRNEngine(num_classes=C, blocks=28, width_factor=10, device=dev, seed=0)
    E.train_step(eng, crit, x, y, 0.01)
    rep = eng.cu_share_report
    row = []
    for name, setup in (("calibrated default", None), ("no sharing", lambda: eng.set_cu_share(None)),
                        ("split forced", lambda: eng.set_cu_share(47.0, calibrate=False))):
        if setup is not None:
            setup()
        for _ in range(3):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        row.append(f"{name} {1e3 * (time.perf_counter() - t0) / 20:7.3f} ms")
    print(f"B={B} C={C}: calibration {rep}\n      " + "   ".join(row), flush=True)
    del eng
