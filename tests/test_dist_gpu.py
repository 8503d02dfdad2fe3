"""The RCCL leg of the data-parallel path on real hardware.  A 1-GPU box cannot run world_size 2, so
this drives the exact code path of a multi-GPU step (process-group init with backend "nccl" = RCCL,
bucketed all-reduce of the flat gradient on the side stream, stream ordering back into SGD) in a
1-rank group and checks it changes nothing; the 2-rank arithmetic is covered on CPU by
tests/test_dist.py (gloo)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.nn as nn

pytestmark = pytest.mark.gpu

from nbdt import dist as ndist  # noqa: E402
from nbdt.engine import WRNEngine, train_step  # noqa: E402
from nbdt.loss import SoftTreeSupLoss  # noqa: E402


def test_rccl_bucketed_allreduce_in_a_one_rank_group():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(),
                               hierarchy="induced-wrn28_10_cifar10")
        g = torch.Generator().manual_seed(0)
        x = torch.randn(16, 3, 32, 32, generator=g).cuda()
        y = torch.randint(0, 10, (16,), generator=g).cuda()
        losses = {}
        for tag, comm in (("plain", None), ("rccl", ndist.GradComm(force=True))):
            eng = WRNEngine(num_classes=10, blocks=10, width_factor=2, device="cuda:0", seed=3)
            losses[tag] = [train_step(eng, crit, x, y, lr=0.05, comm=comm).item() for _ in range(4)]
            losses[tag + "_p"] = eng.store.flat.clone()
        # a 1-rank sum all-reduce is the identity: same trajectory up to the atomics' run-to-run noise
        for a, b in zip(losses["plain"], losses["rccl"]):
            assert abs(a - b) < 2e-2 * abs(a), (losses["plain"], losses["rccl"])
        rel = (losses["plain_p"] - losses["rccl_p"]).norm() / losses["plain_p"].norm()
        assert rel.item() < 3e-2
        # raw collective on a slice of a flat buffer, issued from the side stream
        flat = torch.arange(1024, dtype=torch.float32, device="cuda")
        comm = ndist.GradComm(force=True)
        comm.reduce_range(flat, 128, 512)
        comm.finish(flat)
        assert torch.equal(flat.cpu(), torch.arange(1024, dtype=torch.float32))
    finally:
        dist.destroy_process_group()
