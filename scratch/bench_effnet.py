"""EfficientNet-B0 + SoftTreeSupLoss training-step throughput (BASELINE config 5 shape: 224x224, 1000 classes)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt.engine import train_step
from nbdt.engine_effnet import EfficientNetEngine
from nbdt.loss import SoftTreeSupLoss

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--size", type=int, default=224)
a = ap.parse_args()
eng = EfficientNetEngine(num_classes=1000, device="cuda:0")
crit = SoftTreeSupLoss(dataset="Imagenet1000", criterion=nn.CrossEntropyLoss(), hierarchy="induced-efficientnet_b7b")
g = torch.Generator().manual_seed(0)
x = torch.randn(a.batch, 3, a.size, a.size, generator=g).cuda()
y = torch.randint(0, 1000, (a.batch,), generator=g).cuda()
for _ in range(a.warmup):
    train_step(eng, crit, x, y, lr=0.01)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    loss = train_step(eng, crit, x, y, lr=0.01)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"workload": "EfficientNet-B0 + SoftTreeSupLoss train step", "batch": a.batch, "size": a.size,
                  "ms_per_step": dt * 1e3, "img_per_s": a.batch / dt, "loss": loss.item(),
                  "mem_GB": torch.cuda.max_memory_allocated() / 2**30}))
