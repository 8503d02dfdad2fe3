"""Does a lower-priority weight-gradient stream help?  NBDT_SIDE_PRIO / NBDT_MAIN_PRIO are read here only (experiment)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch
print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
import nbdt.engine as E
from nbdt.engine import WRNEngine, train_step
from nbdt.loss import SoftTreeSupLoss
from nbdt.tree import Tree
dev = torch.device("cuda:0")
def run(side_prio, main_prio, steps=30, warm=5):
    eng = WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
    if side_prio is not None:
        eng._side = torch.cuda.Stream(device=dev, priority=side_prio)
        eng.set_overlap(True)
    crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=torch.nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
    x = torch.randn(512, 3, 32, 32, device=dev); y = torch.randint(0, 10, (512,), device=dev)
    main = torch.cuda.Stream(device=dev, priority=main_prio) if main_prio is not None else torch.cuda.current_stream()
    main.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(main):
        for _ in range(warm): train_step(eng, crit, x, y, lr=0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): train_step(eng, crit, x, y, lr=0.01)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    return dt * 1e3
def run2(late, steps=30, warm=5):
    eng = WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
    eng.wgrad_after_dgrad = late
    crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=torch.nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
    x = torch.randn(512, 3, 32, 32, device=dev); y = torch.randint(0, 10, (512,), device=dev)
    for _ in range(warm): train_step(eng, crit, x, y, lr=0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): loss = train_step(eng, crit, x, y, lr=0.01)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, float(loss)
for rep in range(3):
    for late in (False, True):
        print("weight gradient issued", "after" if late else "before", "its data gradient: %.3f ms/step  loss %.4f" % run2(late))
