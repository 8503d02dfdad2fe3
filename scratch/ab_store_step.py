#!/usr/bin/env python3
"""Training steps of configs 1, 3, 4 with the K-split weight gradient's store + fold epilogue off / on."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E, ops
from nbdt.loss import SoftTreeSupLoss
dev = torch.device("cuda", 0)
CASES = {"c1": ("CIFAR10", "induced-ResNet18", 10, 32, 1.0, 128, "resnet"), "c4": ("TinyImagenet200", "induced-ResNet18", 200, 64, 10.0, 128, "resnet"),
         "c3": ("CIFAR100", "induced-wrn28_10_cifar100", 100, 32, 1.0, 256, "wrn")}
for case in sys.argv[1:] or ["c1", "c4", "c3"]:
    dataset, hier, C, size, tsw, B, kind = CASES[case]
    crit = SoftTreeSupLoss(dataset=dataset, criterion=nn.CrossEntropyLoss(), hierarchy=hier, tree_supervision_weight=tsw)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, size, size, generator=g).to(dev)
    y = torch.randint(0, C, (B,), generator=g).to(dev)
    eng = E.ResNetEngine(C, device=dev) if kind == "resnet" else E.WRNEngine(C, device=dev)
    for _ in range(5):
        E.train_step(eng, crit, x, y, 0.01)
    for r in range(3):
        for v in (0, 1):
            eng.join_side_stream()
            ops.set_wgrad_store_epilogue(v)
            for _ in range(3):
                E.train_step(eng, crit, x, y, 0.01)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                E.train_step(eng, crit, x, y, 0.01)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 30
            print(f"{case} round {r}  store={v}  {ms:7.3f} ms/step  {B / ms * 1e3:8.0f} img/s", flush=True)
    ops.set_wgrad_store_epilogue(0)
