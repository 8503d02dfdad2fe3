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
    # a schedule decision must be rank 0's on every rank (engine.calibrate_cu_share under data parallelism):
    # rank 0 measured "keep", rank 1 measured "discard" -> both keep; and the other way round -> both discard
    assert comm.broadcast_flag(rank == 0, torch.device("cpu")) is True
    assert comm.broadcast_flag(rank != 0, torch.device("cpu")) is False
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


# ------------------------------------------------------------------------------------------------------------
# A whole data-parallel training step on two gloo ranks: each rank runs the fp32 oracle backbone
# + oracle SoftTreeSupLoss on ITS shard, the flat gradient goes through GradComm in the engine's bucket order,
# SGD applies grad/world.  Expected: exactly the sequential computation "each shard through the oracle, average
# the gradients, one SGD step" -- per-shard BatchNorm statistics included (DataParallel semantics, no SyncBN).

def _flat(tensors):
    return torch.cat([t.reshape(-1) for t in tensors])


def _oracle_shard_grads(seed, x, y, pkg):
    nbdt_path.add(oracle=True)
    import nbdt_oracle as O
    import torch_models as TM
    torch.manual_seed(seed)
    net = TM.WRN(10, 10, 1)
    net.train()
    otree = O.OracleTree(*O.default_paths("CIFAR10", "induced-wrn28_10_cifar10", pkg))
    z = net(x)
    loss, dz = O.soft_tree_sup_loss(otree, z.detach().numpy(), y.numpy())
    z.backward(torch.from_numpy(dz))
    params = [p for _, p in net.named_parameters()]
    return net, params, _flat([p.grad for p in params]), float(loss)


def _dp_step_worker(rank, world, port, out_dir, pkg):
    nbdt_path.add()
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.set_num_threads(2)
    from nbdt import dist as ndist
    ndist.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(7)
    gx, gy = torch.randn(8, 3, 32, 32, generator=g), torch.randint(0, 10, (8,), generator=g)
    x, y = ndist.shard_batch(gx, rank, world), ndist.shard_batch(gy, rank, world)
    net, params, flat, _ = _oracle_shard_grads(11, x, y, pkg)       # same seed: identical replicas
    comm = ndist.GradComm()
    n = flat.numel()
    for lo, hi in [(2 * n // 3, n), (n // 3, 2 * n // 3), (0, n // 3)]:   # backward-completion order
        comm.reduce_range(flat, lo, hi)
    comm.finish(flat)
    lr, scale = 0.1, 1.0 / comm.world_size
    with torch.no_grad():
        off = 0
        for p in params:
            p -= lr * scale * flat[off:off + p.numel()].view_as(p)
            off += p.numel()
    np.save(os.path.join(out_dir, f"params{rank}.npy"), _flat([p.detach() for p in params]).numpy())
    torch.distributed.destroy_process_group()


def test_data_parallel_step_equals_per_shard_oracle_average(tmp_path, pkg_dir):
    world, port = 2, _free_port()
    mp.spawn(_dp_step_worker, args=(world, port, str(tmp_path), pkg_dir), nprocs=world, join=True)
    g = torch.Generator().manual_seed(7)
    gx, gy = torch.randn(8, 3, 32, 32, generator=g), torch.randint(0, 10, (8,), generator=g)
    grads = []
    for r in range(world):
        net, params, flat, _ = _oracle_shard_grads(11, gx[4 * r:4 * r + 4], gy[4 * r:4 * r + 4], pkg_dir)
        grads.append(flat)
    torch.manual_seed(11)
    expect = _flat([p.detach() for p in params]) - 0.1 * (grads[0] + grads[1]) / 2     # params: same init on every rank
    got = [np.load(tmp_path / f"params{r}.npy") for r in range(world)]
    assert np.array_equal(got[0], got[1])                      # replicas stay bit-identical
    np.testing.assert_allclose(got[0], expect.numpy(), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------------------
# RCCL's footprint is bounded by this package, not left to RCCL's tuner: the channel count
# is set before the communicator is created and reported; the exposed-all-reduce model is plain arithmetic.

def test_rccl_channel_bound_and_allreduce_model(monkeypatch):
    nbdt_path.add()
    from nbdt import dist as ndist
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.delenv("NBDT_RCCL_CHANNELS", raising=False)
    ndist._bound_rccl_footprint()
    assert os.environ["NCCL_MAX_NCHANNELS"] == "8" and ndist.rccl_channels() == 8
    assert ndist.RCCL["set_by"] == "nbdt.dist default"
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    monkeypatch.setenv("NBDT_RCCL_CHANNELS", "16")
    ndist._bound_rccl_footprint()
    assert os.environ["NCCL_MAX_NCHANNELS"] == "16" and ndist.rccl_channels() == 16
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "4")              # an explicit RCCL setting wins
    ndist._bound_rccl_footprint()
    assert ndist.rccl_channels() == 4 and "environment" in ndist.RCCL["set_by"]
    # ring all-reduce: 2 (N-1)/N of the buffer per rank at channels x 20 GB/s
    nbytes = 146_000_000
    assert ndist.allreduce_model_ms(nbytes, 1) == 0.0
    assert abs(ndist.allreduce_model_ms(nbytes, 8, channels=8) - 1e3 * 1.75 * nbytes / 160e9) < 1e-9
    assert abs(ndist.allreduce_model_ms(nbytes, 2, channels=8) - 1e3 * nbytes / 160e9) < 1e-9
    # the knobs of ADVICE r4: opt-out, a minimum above the bound, reservation from the bound actually in effect
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    monkeypatch.setenv("NBDT_RCCL_CHANNELS", "0")              # leave RCCL alone: nothing exported, footprint unknown
    ndist._bound_rccl_footprint()
    assert "NCCL_MAX_NCHANNELS" not in os.environ and ndist.rccl_channels() is None
    assert ndist.reserved_cus_for_rccl() == ndist.UNKNOWN_CHANNELS_RESERVE
    monkeypatch.setenv("NBDT_RCCL_RESERVED_CUS", "24")
    assert ndist.reserved_cus_for_rccl() == 24
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "12")             # a group created elsewhere under the caller's own bound
    assert ndist.reserved_cus_for_rccl() == 12
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    monkeypatch.delenv("NBDT_RCCL_CHANNELS")
    monkeypatch.setenv("NCCL_MIN_NCHANNELS", "12")             # RCCL would ignore a bound below its minimum
    ndist._bound_rccl_footprint()
    assert os.environ["NCCL_MAX_NCHANNELS"] == "12" and ndist.reserved_cus_for_rccl() == 12
    assert "NCCL_MIN_NCHANNELS" in ndist.RCCL["set_by"]
    monkeypatch.delenv("NCCL_MIN_NCHANNELS")
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    ndist.RCCL.update(max_nchannels=None, set_by=None)
    # a gloo / single-process GradComm holds no CUs
    comm = ndist.GradComm()
    assert comm.reserved_cus == 0 and comm.describe(nbytes)["allreduce_model_ms"] == 0.0
