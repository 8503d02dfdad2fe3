"""One BASELINE.json configuration other than the headline one, for N steps (to be wrapped in rocprofv3).
usage: python scratch/run_config.py <c1|c1b|c3|c4|c4inf|c4inf1024|c5> [--steps N] [--warmup W] [--one-stream] [--batch B]
Prints one JSON line {config, batch, ms, img_per_s}."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt.engine import ResNetEngine, WRNEngine, train_step
from nbdt.engine_effnet import EfficientNetEngine
from nbdt.loss import SoftTreeSupLoss
from nbdt.model import HardEmbeddedDecisionRules
from nbdt.tree import Tree

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--one-stream", action="store_true")
ap.add_argument("--set", action="append", default=[], help="engine attribute override, e.g. fuse_bn1_bwd=0")
a = ap.parse_args()
DEV = "cuda:0"
CASES = {   # engine factory, dataset, hierarchy, batch, image size, classes, tree-supervision weight
    "c1": (lambda: ResNetEngine(10, device=DEV), "CIFAR10", "induced-ResNet18", 128, 32, 10, 1.0),
    "c1b": (lambda: ResNetEngine(10, device=DEV), "CIFAR10", "induced-ResNet18", 512, 32, 10, 1.0),
    "c3": (lambda: WRNEngine(100, device=DEV), "CIFAR100", "induced-wrn28_10_cifar100", 256, 32, 100, 1.0),
    "c4": (lambda: ResNetEngine(200, device=DEV), "TinyImagenet200", "induced-ResNet18", 128, 64, 200, 10.0),
    "c5": (lambda: EfficientNetEngine(1000, device=DEV), "Imagenet1000", "induced-efficientnet_b7b", 128, 224, 1000, 1.0),
}
name = a.config
inference = name.startswith("c4inf")
mk, dataset, hierarchy, B, size, C, tsw = CASES["c4" if inference else name]
if name == "c4inf1024":
    B = 1024
if a.batch:
    B = a.batch
eng = mk()
if a.one_stream:
    eng.set_overlap(False)
for kv in a.set:
    k, v = kv.split("=")
    assert hasattr(eng, k), k
    setattr(eng, k, type(getattr(eng, k))(int(v)))
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, size, size, generator=g).to(DEV)
y = torch.randint(0, C, (B,), generator=g).to(DEV)
if inference:
    rules = HardEmbeddedDecisionRules(tree=Tree(dataset, hierarchy=hierarchy))
    fn = lambda: rules.predict(eng.forward(x, training=False))
else:
    crit = SoftTreeSupLoss(dataset=dataset, criterion=nn.CrossEntropyLoss(), hierarchy=hierarchy,
                           tree_supervision_weight=tsw)
    fn = lambda: train_step(eng, crit, x, y, 0.01)
for _ in range(a.warmup):
    fn()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    fn()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"config": name, "batch": B, "one_stream": a.one_stream, "ms": round(dt * 1e3, 3),
                  "img_per_s": round(B / dt, 1)}))
