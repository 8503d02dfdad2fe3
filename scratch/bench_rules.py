"""Wall time of the rules-layer kernels at the BASELINE shapes (HIP events, 200 launches each), next to the
numpy oracle on the host and the algorithmic bytes (soft fwd 8*C B/sample, fused loss 8*C+12)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add(oracle=True)
import numpy as np, torch
import nbdt_oracle as O
from nbdt import _C
from nbdt.tree import Tree
DEV = "cuda:0"
PKG = os.path.join(nbdt_path.PKG_DIR, "nbdt")
SHAPES = [(512, "CIFAR10", "induced-wrn28_10_cifar10"), (1024, "CIFAR100", "induced-wrn28_10_cifar100"),
          (1024, "TinyImagenet200", "induced-ResNet18"), (256, "Imagenet1000", "induced-efficientnet_b7b")]


def gpu_us(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for B, ds, h in SHAPES:
    tree = Tree(ds, hierarchy=h); otree = O.OracleTree(*O.default_paths(ds, h, PKG))
    C = len(tree.classes); hd = tree.device_handle(0)
    z = torch.randn(B, C, device=DEV) * 3; y = torch.randint(0, C, (B,), device=DEV)
    zc, yc = z.cpu().numpy(), y.cpu().numpy()
    row = {"shape": [B, C], "inner_nodes": len(tree.inodes),
           "soft_fwd_us": round(gpu_us(lambda: _C.soft_forward(hd, z)), 2),
           "soft_loss_fwd_bwd_us": round(gpu_us(lambda: _C.soft_tree_loss(hd, z, y, 1.0, 1.0)), 2),
           "hard_loss_fwd_bwd_us": round(gpu_us(lambda: _C.hard_tree_loss(hd, z, y, 1.0, 1.0)), 2),
           "hard_fwd_us": round(gpu_us(lambda: _C.hard_forward(hd, z, want_onehot=False)), 2),
           "algorithmic_bytes_soft_fwd": 8 * C * B, "algorithmic_bytes_loss": (8 * C + 12) * B}
    t0 = time.perf_counter(); O.soft_forward(otree, zc); row["oracle_soft_fwd_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    t0 = time.perf_counter(); O.soft_tree_sup_loss(otree, zc, yc); row["oracle_soft_loss_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    row["soft_fwd_GBps"] = round(row["algorithmic_bytes_soft_fwd"] / row["soft_fwd_us"] / 1e3, 2)
    print(json.dumps(row))
