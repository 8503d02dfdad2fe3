"""N>1 path on CPU: two gloo processes exercise the host logic of the data-parallel step --
batch sharding, stage-bucketed all-reduce of the flat gradient buffer, averaging via grad_scale."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

import nbdt_path


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    nbdt_path.add()
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from nbdt import dist as ndist
    r, w, _ = ndist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # global batch sharded contiguously along dim 0
    gb = torch.arange(8 * 3).view(8, 3)
    shard = ndist.shard_batch(gb, r, w)
    assert shard.shape[0] == 4 and shard[0, 0].item() == rank * 12
    # flat "gradient" buffer, reduced in three buckets in backward-completion order
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(1000, generator=g)
    comm = ndist.GradComm()
    assert comm.world_size == world
    for lo, hi in [(600, 1000), (250, 600), (0, 250)]:
        comm.reduce_range(flat, lo, hi)
    comm.finish(flat)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), flat.numpy())
    torch.distributed.destroy_process_group()


def test_bucketed_allreduce_two_ranks(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    expect = sum(torch.randn(1000, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        np.testing.assert_allclose(got, expect.numpy(), rtol=1e-6, atol=1e-6)


def test_single_process_is_a_noop():
    nbdt_path.add()
    from nbdt import dist as ndist
    comm = ndist.GradComm()
    flat = torch.ones(10)
    comm.all_reduce_grads(flat)
    assert comm.world_size == 1 and torch.equal(flat, torch.ones(10))
