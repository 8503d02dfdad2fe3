"""nbdt_dwconv_bwd_weight at the EfficientNet-B0 depthwise layers (batch 128, 224x224 input), HIP events over 20 calls,
with its algorithmic bytes (x and gy read once) and a torch fp32 check of the result."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "neural-backed-decision-trees_amd"))
import torch
import torch.nn.functional as F
from nbdt import ops

DEV = "cuda:0"
LAYERS = [(32, 3, 1, 112), (96, 3, 2, 112), (144, 3, 1, 56), (144, 5, 2, 56), (240, 5, 1, 28), (240, 3, 2, 28),
          (480, 3, 1, 14), (480, 5, 1, 14), (672, 5, 1, 14), (672, 5, 2, 14), (1152, 5, 1, 7), (1152, 3, 1, 7)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
if len(sys.argv) > 2 and sys.argv[2] == "det":
    ops.set_deterministic(True)
tot = 0.0
for C, k, s, H in LAYERS:
    Ho = H // s
    g = torch.Generator().manual_seed(C + k)
    xi = torch.randn(B, H, H, C, generator=g)
    gi = torch.randn(B, Ho, Ho, C, generator=g)
    x, gy = ops.padded(B, H, H, C, DEV), ops.padded(B, Ho, Ho, C, DEV)
    ops.interior(x).copy_(xi.to(DEV))
    ops.interior(gy).copy_(gi.to(DEV))
    dw = torch.zeros(k * k, C, device=DEV)
    ops.dwconv_bwd_weight(x, gy, dw, k, s)
    # reference: depthwise conv weight gradient in fp32 on the bf16-rounded operands
    xr = ops.interior(x).float().permute(0, 3, 1, 2).contiguous().requires_grad_(False)
    gr = ops.interior(gy).float().permute(0, 3, 1, 2).contiguous()
    w = torch.zeros(C, 1, k, k, device=DEV, requires_grad=True)
    if s == 1:
        y = F.conv2d(xr, w, padding=k // 2, groups=C)
    else:   # 'same' padding of a stride-2 conv on an even size: (k-2)//2 before ... handled like the engine: pad k//2, stride 2
        y = F.conv2d(xr, w, padding=k // 2, stride=2, groups=C)[:, :, :Ho, :Ho]
    y.backward(gr)
    ref = w.grad[:, 0].permute(1, 2, 0).reshape(k * k, C)
    rel = ((dw - ref).norm() / ref.norm()).item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.dwconv_bwd_weight(x, gy, dw, k, s)
    e0.record()
    for _ in range(20):
        ops.dwconv_bwd_weight(x, gy, dw, k, s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = (x.numel() + gy.numel()) * 2 / 1e6
    tot += us
    print(f"C={C:5d} k={k} s={s} {H:3d}x{H:<3d}: {us:7.1f} us  {mb:7.1f} MB  {mb / us:5.2f} TB/s   rel err {rel:.1e}", flush=True)
print(f"sum {tot:.1f} us")
