#!/usr/bin/env python3
"""K-split weight gradient: plain stores + fold against fp32 atomics -- each WRN shape alone, then the whole step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E, ops
from nbdt.loss import SoftTreeSupLoss
dev = torch.device("cuda", 0)
DEV = "cuda:0"
SHAPES = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(512, 32, 160), (512, 16, 320), (512, 8, 640)]
for (B, H, C) in SHAPES:
    x = ops.padded(B, H, H, C, DEV); ops.interior(x).normal_()
    g = ops.padded(B, H, H, C, DEV); ops.interior(g).normal_()
    d = ops.conv_wgrad_desc(B, H, H, C, C, 3, 1)
    out = {}
    for mode in (0, 1, 0, 1):
        ops.set_wgrad_store_epilogue(mode)
        dw = torch.zeros(C, 9, C, device=DEV)
        for _ in range(5): ops.conv_wgrad(d, x, g, dw)
        dw.zero_(); ops.conv_wgrad(d, x, g, dw); out[mode] = dw.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): ops.conv_wgrad(d, x, g, dw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        fl = 2.0 * B * H * H * C * C * 9
        print(f"B={B} {H}x{H} C={C} store={mode}: {us:7.1f} us/launch (with the fold)  {fl / us * 1e-6:6.0f} TF/s", flush=True)
    rel = ((out[0] - out[1]).norm() / out[0].norm()).item()
    ops.set_wgrad_store_epilogue(1)
    dw2 = torch.zeros(C, 9, C, device=DEV); ops.conv_wgrad(d, x, g, dw2)
    print(f"   store vs atomic rel-L2 {rel:.2e}; store twice bit-identical: {bool((dw2 == out[1]).all())}", flush=True)

if len(sys.argv) > 1:
    sys.exit(0)
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
gen = torch.Generator().manual_seed(0)
x = torch.randn(512, 3, 32, 32, generator=gen).to(dev)
y = torch.randint(0, 10, (512,), generator=gen).to(dev)
eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
eng.set_cu_share(47.0, calibrate=False)
for _ in range(5):
    E.train_step(eng, crit, x, y, 0.01)
for r in range(3):
    for v in (0, 1):
        eng.join_side_stream()
        ops.set_wgrad_store_epilogue(v)
        for _ in range(4):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 30
        print(f"round {r}  store={v}  {ms:7.3f} ms/step  {512 / ms * 1e3:8.0f} img/s", flush=True)
