#!/usr/bin/env python3
"""A/B of engine.set_cu_share on the bench workload (WRN-28-10, B=512): ms/step with the BatchNorm-backward passes
confined to n CUs beside the weight gradients, for a list of settings, alternating, one process.
usage: cu_share_ab.py [--steps 40] [--rounds 2] setting ...   setting = off | gbps:target_us:min:max[:join][:split<target_us>]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path  # noqa: E402

nbdt_path.add()
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from nbdt import engine as E  # noqa: E402
from nbdt.loss import SoftTreeSupLoss  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("settings", nargs="+")
args = ap.parse_args()
dev = torch.device("cuda", 0)
eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(1234)
img = torch.randn(512, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (512,), generator=g).to(dev)


def apply(setting):
    if setting == "off":
        eng.set_cu_share(None)
        return
    f = setting.split(":")
    flags = f[4] if len(f) > 4 else ""
    split_us = [float(x[5:]) for x in flags.split("-") if x.startswith("split")]
    eng.set_cu_share(float(f[0]), float(f[1]), int(f[2]), int(f[3]), join=("join" in flags), calibrate=False,
                     split_reduce=bool(split_us), split_target_us=split_us[0] if split_us else 250.0)


# gradients of one step: share on vs off (same weights, same batch): equal up to the order of fp32 atomics
def grads(setting):
    apply(setting)
    eng.zero_grad()
    z = eng.forward(img, training=True)
    loss, gz = crit.loss_and_grad(z, y)
    eng.backward(gz)
    torch.cuda.synchronize()
    return eng.store.grad.clone()


g_off = grads("off")
for s in args.settings:
    if s != "off":
        g_on = grads(s)
        rel = ((g_on - g_off).norm() / g_off.norm()).item()
        g_off2 = grads("off")
        rel0 = ((g_off2 - g_off).norm() / g_off.norm()).item()
        # (two identical launches differ by ~0.2-0.3: the order of the statistics' atomics moves 1-ulp bf16 roundings,
        # which flip ReLU masks; exactness of the two kernels is checked in scratch/cu_share_check.py / the tests)
        print(f"gradient check {s}: rel-L2 difference to 'off' {rel:.3e}; 'off' to 'off' {rel0:.3e}", flush=True)
        break

for _ in range(5):
    E.train_step(eng, crit, img, y, 0.01)
for r in range(args.rounds):
    for s in args.settings:
        apply(s)
        for _ in range(3):
            E.train_step(eng, crit, img, y, 0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            E.train_step(eng, crit, img, y, 0.01)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        print(f"round {r} {s:>24s}: {ms:7.3f} ms/step  {512e3 / ms:8.0f} img/s", flush=True)
