#!/usr/bin/env python3
"""What does a third party holding CUs (an RCCL all-reduce kernel) do to the training step on ONE GPU?
The MFMA kernels are one persistent block per CU and the CU-sharing schedule plans with every CU of every XCD, so a
collective's kernel holding k CUs during backward delays exactly the blocks that were meant for them.  This emulates
the exchange of a data-parallel step: a GradComm look-alike whose `reduce_range` launches probes/libcu_hold.so's
holder (k blocks x 256 threads, no LDS, for a time proportional to the bucket's bytes, 1.5 ms per step in total) on a
third stream at the points where the real all-reduce buckets are issued.
usage: contention_probe.py [--steps 20] [--total-us 1500]"""
import argparse, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch
import torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--total-us", type=float, default=1500.0)
args = ap.parse_args()
hold = ctypes.CDLL(os.path.join(ROOT, "probes", "libcu_hold.so"))
hold.cu_hold.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
dev = torch.device("cuda", 0)


class HolderComm:
    """GradComm's interface; the 'all-reduce' of a bucket is a holder kernel on a third stream."""
    world_size = 1

    def __init__(self, k, total_us, total_elems):
        self.k, self.total_us, self.total = k, total_us, total_elems
        self.stream = torch.cuda.Stream(device=dev)

    def broadcast_flag(self, flag, device):
        return bool(flag)

    def reduce_range(self, flat, lo, hi):
        if self.k == 0:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.stream.wait_event(ev)
        ticks = int(self.total_us * (hi - lo) / self.total * 100)      # 100 MHz wall clock
        assert hold.cu_hold(self.k, ticks, ctypes.c_void_p(self.stream.cuda_stream)) == 0

    def finish(self, flat=None):
        torch.cuda.current_stream(dev).wait_stream(self.stream)


crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(0)
x = torch.randn(512, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (512,), generator=g).to(dev)
eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
modes = {"no sharing": lambda: eng.set_cu_share(None),
         "split sharing": lambda: eng.set_cu_share(47.0, calibrate=False),
         "split sharing + join": lambda: eng.set_cu_share(47.0, calibrate=False, join=True)}
print(f"# ms/step at 512 images, holder of k blocks for {args.total_us:.0f} us per step in three bucket-sized pieces")
for k in (0, 8, 16, 32, 64):
    row = []
    for name, setup in modes.items():
        setup()
        comm = HolderComm(k, args.total_us, eng.store.grad.numel())
        for _ in range(4):
            E.train_step(eng, crit, x, y, 0.01, comm=comm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            E.train_step(eng, crit, x, y, 0.01, comm=comm)
        torch.cuda.synchronize()
        row.append(f"{name}: {1e3 * (time.perf_counter() - t0) / args.steps:7.3f}")
    print(f"k={k:2d}   " + "   ".join(row), flush=True)
