#!/usr/bin/env python3
"""Fused head (nbdt_head_soft_tree_loss) vs the three launches it replaces, HIP-event time per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch
from nbdt import _C, ops
from nbdt._C import lib
from nbdt.tree import Tree
dev = "cuda:0"
def timeit(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for ds, h, B, K in (("CIFAR10", "induced-wrn28_10_cifar10", 512, 640), ("CIFAR100", "induced-wrn28_10_cifar100", 256, 640),
                    ("TinyImagenet200", "induced-ResNet18", 128, 512)):
    tree = Tree(ds, hierarchy=h); handle = tree.device_handle(0); C = tree.flat.num_classes
    g = torch.Generator().manual_seed(0)
    pooled = torch.rand(B, K, generator=g).to(dev); W = (torch.randn(C, K, generator=g) * K ** -0.5).to(dev)
    bias = torch.zeros(C, device=dev); y = torch.randint(0, C, (B,), generator=g).to(dev)
    gW, gb = torch.zeros_like(W), torch.zeros_like(bias)
    z = torch.empty(B, C, device=dev); gp = torch.empty_like(pooled)
    def unfused():
        ops.linear_fwd(pooled, W, bias, z)
        _, gz = _C.soft_tree_loss(handle, z, y, 1.0, 1.0)
        ops.linear_bwd(pooled, W, gz, gp, gW, gb)
    line = f"{ds} B={B} C={C} K={K}: three launches {timeit(unfused):6.1f} us"
    for spb in (16, 8, 4, 1):
        pass  # (the samples-per-block switch was a round-3 experiment build)
        t = timeit(lambda: _C.head_soft_tree_loss(handle, pooled, W, bias, y, 1.0, 1.0, gW=gW, gb=gb))
        line += f" | fused, <= {spb:2d} samples/block {t:6.1f} us"
    print(line, flush=True)
