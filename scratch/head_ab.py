"""Fused head (nbdt_head_soft_tree_loss) against linear_fwd -> soft_tree_loss -> linear_bwd at the BASELINE shapes."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "neural-backed-decision-trees_amd"))
import torch, torch.nn as nn
from nbdt import ops
from nbdt.loss import SoftTreeSupLoss

DEV = "cuda:0"
SHAPES = [("C1 wrn28_10 cifar10", "CIFAR10", "induced-wrn28_10_cifar10", 10, 640, 512, 1.0),
          ("C3 wrn28_10 cifar100", "CIFAR100", "induced-wrn28_10_cifar100", 100, 640, 128, 1.0),
          ("C4 resnet18 tiny200", "TinyImagenet200", "induced-ResNet18", 200, 512, 128, 10.0)]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, ds, hier, C, K, B, tsw in SHAPES:
    crit = SoftTreeSupLoss(dataset=ds, criterion=nn.CrossEntropyLoss(), hierarchy=hier, tree_supervision_weight=tsw)
    g = torch.Generator().manual_seed(0)
    pooled = torch.randn(B, K, generator=g).to(DEV)
    W = (torch.randn(C, K, generator=g) * 0.05).to(DEV)
    b = torch.zeros(C, device=DEV)
    y = torch.randint(0, C, (B,), generator=g).to(DEV)
    gW, gb = torch.zeros_like(W), torch.zeros_like(b)
    z = torch.empty(B, C, device=DEV)
    gx = torch.empty(B, K, device=DEV)

    def fused():
        return crit.head_loss_and_grad(pooled, W, b, y, grad_weight=gW, grad_bias=gb)

    def unfused():
        ops.linear_fwd(pooled, W, b, z)
        loss, gz = crit.loss_and_grad(z, y)
        ops.linear_bwd(pooled, W, gz, gx, gW, gb)
        return loss, gx

    gW.zero_(); gb.zero_()
    l1, g1, _ = fused()
    w1 = gW.clone(); gW.zero_(); gb.zero_()
    l2, g2 = unfused()
    w2 = gW.clone()
    print(f"{name}: B={B} K={K} C={C}  fused {timed(fused):.1f} us   unfused {timed(unfused):.1f} us   "
          f"loss diff {abs(l1.item() - l2.item()):.2e}  gx rel {((g1 - g2).norm() / g2.norm()).item():.2e}  "
          f"dW rel {((w1 - w2).norm() / w2.norm()).item():.2e}", flush=True)
