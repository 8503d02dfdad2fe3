"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

Replaces ``torch.nn.DataParallel`` (reference main.py:160-162: replicate / scatter / gather /
reduce-to-GPU-0 every step).  Here each rank owns a full replica, runs the loss on its own shard of
the minibatch (like the reference's DDP example, examples/imagenet/losses/nbdt_losses.py:6-21), and
the ONLY exchange is a sum all-reduce of the flat fp32 gradient buffer, issued per stage-sized
bucket on a side stream as soon as backward has produced it, so it overlaps the remaining
backward kernels.  The 1/world_size averaging is folded into the SGD kernel (``grad_scale``).

BatchNorm statistics stay per-rank, exactly like DataParallel's per-replica BN (no SyncBN in the
reference).  With the ``gloo`` backend and CPU tensors the same class runs in the unit tests.
"""
import os

import torch
import torch.distributed as dist

# RCCL's footprint on the chip, chosen HERE rather than left to its tuner: an all-reduce kernel is one thread block per
# channel, and every such block holds a CU that a one-block-per-CU MFMA kernel of the backward pass then cannot use.
# 8 channels (one per XCD) move the 146 MB of WRN-28-10 gradients in ~1.7 ms at the ~20 GB/s a channel is ASSUMED to
# sustain over xGMI -- inside the ~11 ms of backward they overlap with; at a third of that rate the exchange still ends
# before backward does, because the largest bucket (stage 3, ~3/4 of the bytes) is ready first -- and cost the MFMA
# kernels 8 of 256 CUs while buckets are in flight (engine.backward -> ops.set_reserved_cus).
#
# The bound is a DEFAULT, not a fact about the fabric: bench.py's `comm` object reports the exposed
# all-reduce time next to this model, and every knob is overridable without touching code:
#   NBDT_RCCL_CHANNELS=n   bound RCCL to n channels (n >= 1); NBDT_RCCL_CHANNELS=0: do not touch RCCL at all
#   NCCL_MAX_NCHANNELS     already in the environment: wins over everything here (and is what the CU reservation uses)
#   NCCL_MIN_NCHANNELS     RCCL raises its channel count to at least this; a minimum above the bound would make RCCL
#                          ignore the bound, so the bound (and the CU reservation) is raised to it
#   NBDT_RCCL_RESERVED_CUS CUs to keep free for the collective when its channel count is unknown (process group created
#                          elsewhere, or NBDT_RCCL_CHANNELS=0); default UNKNOWN_CHANNELS_RESERVE
# The variable caps every RCCL communicator of the process: a host program with other collectives sets its own value.
DEFAULT_RCCL_CHANNELS = 8
UNKNOWN_CHANNELS_RESERVE = 16        # CUs left to a collective whose channel count nobody bounded
ASSUMED_GBPS_PER_CHANNEL = 20.0      # stated assumption until a multi-GPU node measures it (bench.py's `comm` object)
RCCL = {"max_nchannels": None, "set_by": None}


def rccl_channels():
    """Channel bound in effect for RCCL (None: not a RCCL process group / left to RCCL)."""
    return RCCL["max_nchannels"]


def _bound_rccl_footprint():
    env = os.environ
    floor = int(env["NCCL_MIN_NCHANNELS"]) if env.get("NCCL_MIN_NCHANNELS", "").isdigit() else 0
    if "NCCL_MAX_NCHANNELS" in env:
        RCCL.update(max_nchannels=max(int(env["NCCL_MAX_NCHANNELS"]), floor), set_by="NCCL_MAX_NCHANNELS (environment)")
        return
    n = int(env.get("NBDT_RCCL_CHANNELS", DEFAULT_RCCL_CHANNELS))
    if n <= 0:                                          # opt-out: RCCL's tuner decides, the footprint is unknown
        RCCL.update(max_nchannels=None, set_by="NBDT_RCCL_CHANNELS=0 (left to RCCL)")
        return
    by = "NBDT_RCCL_CHANNELS" if "NBDT_RCCL_CHANNELS" in env else "nbdt.dist default"
    if floor > n:
        n, by = floor, by + ", raised to NCCL_MIN_NCHANNELS"
    env["NCCL_MAX_NCHANNELS"] = str(n)                  # read by RCCL when the communicator is created
    RCCL.update(max_nchannels=n, set_by=by)


def reserved_cus_for_rccl():
    """CUs the one-block-per-CU kernels leave to RCCL's kernels: the channel bound IN EFFECT (ours, or an
    NCCL_MAX_NCHANNELS the process group was created under), else a stated reserve for an unbounded communicator."""
    n = rccl_channels()
    if n is None and os.environ.get("NCCL_MAX_NCHANNELS", "").isdigit():
        n = int(os.environ["NCCL_MAX_NCHANNELS"])        # group created elsewhere, under the caller's own bound
    if n is None:
        n = int(os.environ.get("NBDT_RCCL_RESERVED_CUS", UNKNOWN_CHANNELS_RESERVE))
    return max(0, min(int(n), 64))


def allreduce_model_ms(nbytes, world, channels=None, gbps_per_channel=ASSUMED_GBPS_PER_CHANNEL):
    """Time of a ring all-reduce of `nbytes` per rank under the stated per-channel rate: every rank sends and receives
    2 (world - 1) / world of the buffer.  An ASSUMPTION-based model (no peer to measure against on a 1-GPU box); with
    real peers bench.py reports the measured exposure beside it."""
    if world <= 1:
        return 0.0
    channels = channels or rccl_channels() or DEFAULT_RCCL_CHANNELS
    return 1e3 * (2.0 * (world - 1) / world * nbytes) / (channels * gbps_per_channel * 1e9)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world_size, local_rank); a single process without those variables is (0, 1, 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            _bound_rccl_footprint()
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_batch(t, rank, world):
    """Rank's contiguous shard of a global batch along dim 0 (global batch must divide evenly)."""
    n = t.shape[0]
    if n % world != 0:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return t[rank * per:(rank + 1) * per]


class GradComm:
    """Bucketed, overlapped sum all-reduce of a flat gradient buffer."""

    def __init__(self, group=None, force=False):
        """force=True issues the collectives even in a 1-rank group (exercises the RCCL path on a
        single-GPU box; see tests/test_dist_gpu.py)."""
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = bool(force) and dist.is_initialized()
        self._stream = None
        self._work = []
        self.last_ranges = []     # (lo, hi) element ranges of the buckets issued since the last finish(), in issue order
        self._issued = []
        # CUs the collective's kernels occupy while a bucket is in flight (one block per channel); 0 for gloo / 1 rank
        rccl = dist.is_initialized() and dist.get_backend(group) == "nccl" and (self.world_size > 1 or self.force)
        self.reserved_cus = reserved_cus_for_rccl() if rccl else 0

    def describe(self, nbytes):
        """What bench.py puts into its `comm` object about the exchange of `nbytes` per rank."""
        return {"bucket_ranges_last_step": [list(r) for r in self.last_ranges],
                "bucket_bytes_last_step": [4 * (hi - lo) for lo, hi in self.last_ranges],
                "rccl_max_nchannels": rccl_channels(), "rccl_channels_set_by": RCCL["set_by"],
                "reserved_cus_while_buckets_in_flight": self.reserved_cus,
                "allreduce_model_ms": round(allreduce_model_ms(nbytes, self.world_size), 3),
                "allreduce_model": f"ring, 2(N-1)/N x {nbytes} B per rank over {rccl_channels() or DEFAULT_RCCL_CHANNELS} "
                                   f"channels x {ASSUMED_GBPS_PER_CHANNEL} GB/s (assumed per-channel xGMI rate)"}

    def reduce_range(self, flat, lo, hi):
        """Launch the all-reduce of flat[lo:hi]; everything already enqueued on the current stream
        that wrote this range is ordered before it."""
        if (self.world_size == 1 and not self.force) or hi <= lo:
            return
        self._issued.append((int(lo), int(hi)))
        view = flat[lo:hi]
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(ev)
                self._work.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._work.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def broadcast_flag(self, flag, device):
        """Rank 0's boolean, adopted by every rank (schedule decisions that must not differ between replicas)."""
        if self.world_size == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        return bool(t.item())

    def finish(self, flat=None):
        """Order every outstanding all-reduce before subsequent work on the current stream."""
        for w in self._work:
            w.wait()
        self._work = []
        self.last_ranges, self._issued = self._issued, []
        if self._stream is not None and flat is not None and flat.is_cuda:
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)

    def all_reduce_grads(self, flat, buckets=None):
        """Whole-buffer convenience form (used when backward did not stream buckets itself)."""
        if self.world_size == 1:
            return
        for lo, hi in (buckets or [(0, flat.numel())]):
            self.reduce_range(flat, lo, hi)
        self.finish(flat)
