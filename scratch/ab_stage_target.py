#!/usr/bin/env python3
"""Same-box A/B of the CU split per stage (engine.share_stage_us: output-grid pixels -> split_target_us): one stage
varied at a time, the others at the shipped 190 us.  Prints the (weight-gradient budget, pass CUs) each value gives.
WRN-28-10, 512 images.   usage: ab_stage_target.py [--steps 30] [--rounds 2]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E, ops
from nbdt.loss import SoftTreeSupLoss
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()
dev = torch.device("cuda", 0)
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(0)
x = torch.randn(args.batch, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (args.batch,), generator=g).to(dev)
eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
eng.set_cu_share(47.0, calibrate=False)
if os.environ.get("AB_NO_BATCH_SCALE"):
    eng.share_scale_batch = False
for _ in range(5):
    E.train_step(eng, crit, x, y, 0.01)
B = args.batch
# what each value means in CUs (bn2 of a dense unit of the stage)
for hw, c in ((1024, 160), (256, 320), (64, 640)):
    u = [u for u in eng.units if u["cout"] == c and u["idconv"] is None][0]
    side = int(hw ** 0.5)
    desc = u["conv2"].plan(B, side, side)[3]
    for us in (110, 130, 150, 170, 190, 210, 240, 280):
        b2, n2 = ops.plan_cu_share(desc, B * hw * c, eng.share_bn2_tensors, 47.0, us, 16, 128)
        b1, n1 = ops.plan_cu_share(desc, B * hw * c, eng.share_bn1_tensors, 47.0, us, 16, 128)
        print(f"# grid {hw:4d} target {us:3d} us: bn2 pass {n2:3d} CUs (budget {b2}), bn1 pass {n1:3d} CUs (budget {b1})")
ap2 = os.environ.get("AB_STAGE_CONFIGS")       # e.g. "{};{1024:170,256:150}": explicit list instead of the one-stage sweep
if ap2:
    import ast
    configs = [ast.literal_eval(c) for c in ap2.split(";")]
else:
    configs = [{}]
    for hw in (1024, 256, 64):
        for us in (110.0, 130.0, 150.0, 170.0, 210.0, 240.0, 280.0):
            configs.append({hw: us})
for r in range(args.rounds):
    for cfg in configs:
        eng.share_stage_us = dict(cfg)
        for _ in range(3):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            E.train_step(eng, crit, x, y, 0.01)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        print(f"round {r}  share_stage_us={cfg!s:16s}  {ms:7.3f} ms/step  {args.batch / ms * 1e3:8.0f} img/s", flush=True)
