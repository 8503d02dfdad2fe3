"""Throughput of the BASELINE.json configs other than the headline one (single MI355X, synthetic data).
C1 ResNet18/CIFAR10 SoftTreeSupLoss train B=128; C3 WRN-28-10/CIFAR100 SoftTreeSupLoss train B=256 (per-GPU
share of the 4-GPU config); C4 ResNet18/TinyImagenet200 64x64: HardNBDT inference + SoftTreeSupLoss(tsw 10)
train, B=128; C5 EfficientNet-B0/Imagenet1000 224x224 SoftTreeSupLoss train B=128."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt.engine import ResNetEngine, WRNEngine, train_step
from nbdt.engine_effnet import EfficientNetEngine
from nbdt.loss import SoftTreeSupLoss
from nbdt.model import HardEmbeddedDecisionRules
from nbdt.tree import Tree

DEV = "cuda:0"


def timeit(fn, steps=15, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def train_case(name, eng, dataset, hierarchy, B, size, C, tsw=1.0):
    crit = SoftTreeSupLoss(dataset=dataset, criterion=nn.CrossEntropyLoss(), hierarchy=hierarchy,
                           tree_supervision_weight=tsw)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, size, size, generator=g).to(DEV); y = torch.randint(0, C, (B,), generator=g).to(DEV)
    dt = timeit(lambda: train_step(eng, crit, x, y, 0.01))
    return {"config": name, "mode": "train step", "batch": B, "ms": round(dt * 1e3, 3), "img_per_s": round(B / dt, 1)}


out = []
out.append(train_case("C1 ResNet18 CIFAR10 SoftTreeSupLoss", ResNetEngine(10, device=DEV), "CIFAR10", "induced-ResNet18", 128, 32, 10))
out.append(train_case("C1' ResNet18 CIFAR10 SoftTreeSupLoss", ResNetEngine(10, device=DEV), "CIFAR10", "induced-ResNet18", 512, 32, 10))
out.append(train_case("C3/GPU WRN-28-10 CIFAR100 SoftTreeSupLoss", WRNEngine(100, device=DEV), "CIFAR100", "induced-wrn28_10_cifar100", 256, 32, 100))
eng = ResNetEngine(200, device=DEV)
out.append(train_case("C4 ResNet18 TinyImagenet200 SoftTreeSupLoss tsw10", eng, "TinyImagenet200", "induced-ResNet18", 128, 64, 200, 10.0))
rules = HardEmbeddedDecisionRules(tree=Tree("TinyImagenet200", hierarchy="induced-ResNet18"))
x = torch.randn(128, 3, 64, 64, device=DEV)
dt = timeit(lambda: rules.predict(eng.forward(x, training=False)))
out.append({"config": "C4 ResNet18 TinyImagenet200 HardNBDT", "mode": "inference (backbone + hard rules)", "batch": 128,
            "ms": round(dt * 1e3, 3), "img_per_s": round(128 / dt, 1)})
x = torch.randn(1024, 3, 64, 64, device=DEV)
dt = timeit(lambda: rules.predict(eng.forward(x, training=False)), steps=8)
out.append({"config": "C4 ResNet18 TinyImagenet200 HardNBDT", "mode": "inference (backbone + hard rules)", "batch": 1024,
            "ms": round(dt * 1e3, 3), "img_per_s": round(1024 / dt, 1)})
del eng
out.append(train_case("C5 EfficientNet-B0 Imagenet1000 SoftTreeSupLoss", EfficientNetEngine(1000, device=DEV), "Imagenet1000",
                      "induced-efficientnet_b7b", 128, 224, 1000))
for o in out:
    print(json.dumps(o))
