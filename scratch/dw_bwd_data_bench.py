"""nbdt_dwconv_bwd_data at EfficientNet-B0's strided depthwise layers (batch 128): HIP events over 20 calls."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "neural-backed-decision-trees_amd"))
import torch
from nbdt import ops
DEV = "cuda:0"
B = 128
for C, k, H in ((96, 3, 112), (144, 5, 56), (240, 3, 28), (672, 5, 14)):
    g = torch.Generator().manual_seed(C)
    gy, gx = ops.padded(B, H // 2, H // 2, C, DEV), ops.padded(B, H, H, C, DEV)
    ops.interior(gy).copy_(torch.randn(B, H // 2, H // 2, C, generator=g).to(DEV))
    w = (torch.randn(k * k, C, generator=g) * 0.3).to(DEV)
    for _ in range(3):
        ops.dwconv_bwd_data(gy, w, gx, k, 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.dwconv_bwd_data(gy, w, gx, k, 2)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = (gy.numel() + gx.numel()) * 2 / 1e6
    print(f"C={C:4d} k={k} {H}x{H} -> gx: {us:7.1f} us  {mb:6.1f} MB  {mb / us:5.2f} TB/s  checksum {gx.float().abs().sum().item():.6e}", flush=True)
