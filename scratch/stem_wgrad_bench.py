#!/usr/bin/env python3
"""Stem weight gradient (3 -> cout, 3x3): time per launch at the three BASELINE stems."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch
from nbdt import ops
DEV = "cuda:0"
for (B, H, cout, cpad, stride, what) in [(512, 32, 16, 32, 1, "WRN-28-10 / CIFAR"), (128, 32, 64, 64, 1, "ResNet18 / CIFAR"),
                                         (128, 64, 64, 64, 1, "ResNet18 / TinyImagenet"), (128, 224, 32, 32, 2, "EfficientNet-B0 / Imagenet")]:
    img = torch.randn(B, 3, H, H, device=DEV)
    Ho = H // stride
    gy = ops.padded(B, Ho, Ho, cpad, DEV); ops.interior(gy).normal_()
    dw = torch.zeros(cout, 27, device=DEV)
    for _ in range(3): ops.stem_wgrad(img, gy, dw, cout, stride=stride)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): ops.stem_wgrad(img, gy, dw, cout, stride=stride)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    mb = (B * Ho * Ho * cpad * 2 + B * 3 * H * H * 4) / 1e6
    print(f"{what:28s} B={B} {H}x{H} cout={cout} stride={stride}: {us:7.1f} us  ({mb:6.1f} MB read once = {mb / us * 1e-3:5.2f} TB/s)", flush=True)
