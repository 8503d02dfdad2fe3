#!/usr/bin/env python3
"""Headline benchmark: images/sec training SoftNBDT WideResNet-28-10 on CIFAR10-shaped synthetic
batches (BASELINE.json `metric`, configs[1]: batch 512 per MI355X, fused rules/loss kernel).

  python bench.py --gpus N --steps K --warmup W
  (N>1 without a torchrun environment: bench.py re-launches itself as
   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...,
   one rank per GPU over RCCL; launched that way by a driver it just reads RANK/LOCAL_RANK/WORLD_SIZE.)

A "step" = nbdt.engine.train_step, the engine's default schedule: zero_grad -> backbone forward to the pooled features
-> classifier + rules + SoftTreeSupLoss forward AND backward in one launch (nbdt_head_soft_tree_loss) -> backbone
backward -> [RCCL all-reduce of gradients, overlapped] -> SGD(momentum .9, wd 5e-4) on one batch already resident in
HBM.  In backward the HBM-bound BatchNorm passes run on ~100 CUs BESIDE the MFMA-bound weight gradients on the other
~160 (engine.set_cu_share, on by default; `cu_share` in the JSON line is the calibration made before the first
backward; --no-cu-share restores the order without it).  Weak scaling: 512 images per GPU.  Prints ONE JSON line on rank 0.

roofline: the dominant kernel is conv_igemm (forward + data-gradient implicit GEMMs; 2/3 of the
step's flops).  achieved = algorithmic flops of its launches / their HIP-event durations, measured
live on the launch stream over min(K, 5) further steps of the same loop right after the K timed steps
(event records between back-to-back kernels would bias `value` by ~5 %), with the engine's second stream
joined so that no other kernel shares the GPU with the measured launch; peak = 2.5 PFLOP/s dense bf16 MFMA
(gfx950).
cpu_baseline: the fp32 CPU oracle port (oracle/torch_models.py WRN + oracle/nbdt_oracle.py loss) on a
bounded sample, all host cores -- test infrastructure used only as the timed baseline here.
agreement ("top-1 vs ref" half of the metric): after the timed loop the trained weights are copied into the
CPU oracle port and both run an eval-mode forward on the same fixed batch; reported are the fraction of equal
argmax(logits), of equal HardNBDT predictions (each path's rules on its own logits) and the logit error.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import nbdt_path  # noqa: E402

nbdt_path.add()

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0
GFLOP_PER_IMG_TRAIN = 31.46     # WRN-28-10 @32x32: fwd 10.487 GFLOP (2*MAC) x3


TRAFFIC_FILES = ("r06_final_hbm_traffic.json", "r05_final_hbm_traffic.json", "r04_final_hbm_traffic.json")   # newest first
LOGIT_TOLERANCE = 3e-2          # max |logit error| / logit scale against the fp32 CPU port


def pmc_traffic(launches_per_step):
    """(bytes, source): HBM-side bytes per launch of the dominant kernel family -- launch-weighted over the
    forward / data-gradient kernels of the schedule this file times (plain-epilogue data gradients) -- from the
    newest committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc runs of this
    same command with --no-overlap; scratch/prof_bench.sh + scratch/traffic_report.py).  bench.py cannot collect PMC counters itself, so the
    number is only quoted when the file was recorded from THE SAME LAUNCHES: its "_meta" entry lists the implicit-GEMM
    kernels' launches per step by device kernel name, and `launches_per_step` is what this run's roofline pass
    counted (nbdt_debug_last_igemm_full after every launch: the device kernel WITH its template arguments, as rocprofv3
    prints it -- round 5's guard compared family names and let a file recorded before a template parameter was added
    through).  Any difference -- a kernel renamed, re-instantiated, added, re-routed -- gives (None, reason) instead of a
    stale number."""
    for fname in TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", fname)
        if os.path.exists(path):
            break
    else:
        return None, "no PMC traffic file under profiles/"
    with open(path) as f:
        t = json.load(f)
    meta = t.pop("_meta", None)
    if meta is None:
        return None, f"profiles/{fname} predates the launch-count record (_meta): not quoted"
    want = {k: round(v, 2) for k, v in meta["igemm_launches_per_step"].items()}
    got = {k: round(v, 2) for k, v in launches_per_step.items()}
    if want != got:
        return None, (f"profiles/{fname} was recorded from other launches (per step, file {want} vs this run {got}): "
                      "not quoted")
    n = b = 0
    for name, v in t.items():
        if "conv3x3_pp_kernel" in name or "conv3x3_halo_kernel" in name or "conv_igemm_dma" in name or "conv_seg_kernel" in name:
            n += v["launches"]
            b += v["launches"] * (v["fetch_bytes_x2"] + v["write_bytes"])
    return (round(b / n) if n else None), "profiles/" + fname


def attainable(dev):
    """What the matrix pipes deliver on THIS box, measured right after the timed loop, so that a
    kernel's distance from the 2.5 PFLOP/s spec splits into power / clock and schedule:

      mfma_stream   register-only v_mfma_f32_32x32x16_bf16 stream, one 512-thread block per CU (two waves per SIMD like the
                    8-wave kernels), per-lane / per-instruction varying operands (csrc/probe.hip): no memory at all --
                    what is missing to 2.5 PF here is power / clock, nothing a schedule can recover;
      attainable    the dominant kernel's OWN K loop in steady state: conv3x3_pp_kernel<5,false,0,8> -- the kernel of the
                    plain data gradients -- on a synthetic 3x3 conv with 2560 input channels (80 K chunks x 9 taps = 720
                    K steps per 512-pixel tile instead of 45-180; 256 tiles, one per CU, operands cache-resident): per K step exactly the production
                    mix (20 MFMA 32x32x16, 14 ds_read_b128, the step's LDS-DMA pieces, the two barriers, the address
                    arithmetic in the MFMA shadow) with exactly the real dependencies, prologue + epilogue < 4 % of the
                    launch.  mfma_stream - attainable = what the loop's own schedule + its memory instructions cost;
                    attainable - achieved = what tile prologues, epilogues and launch boundaries cost."""
    from nbdt import ops
    from nbdt._C import lib, check, ptr
    from nbdt.ops import stream_ptr

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    sink = torch.zeros(4, device=dev)
    blocks, iters = 256, 6000
    t = timed(lambda: check(lib().nbdt_probe_mfma_stream(blocks, iters, ptr(sink), stream_ptr(dev))), 3)
    stream_tf = blocks * 8 * iters * 16 * 32768.0 / t / 1e12
    # 16 images x 32x32, 2560 -> 1280 channels: 32 pixel tiles x 8 cout tiles = 256 items (one per CU), 80 K chunks x 9
    # taps = 720 K steps per tile (production: 45-180); input 95 MB + weights 59 MB stay in the 256 MB Infinity Cache /
    # the L2s like production's operands do (a 1.5 GB input made every halo piece an exposed HBM round trip: 1248 TF/s)
    B, H, cin, cout = 16, 32, 2560, 1280
    x = ops.padded(B, H, H, cin, dev)
    ops.interior(x).normal_()
    wb = (torch.randn(cout, 9, cin, device=dev) * 0.02).to(torch.bfloat16)
    wt = ops.weight_tiles(wb)
    out = ops.padded(B, H, H, cout, dev)
    d = ops.conv_fwd_desc(B, H, H, cin, cout, 3, 1)
    d.w_tiled = wt.data_ptr()
    d.wide_tile = 2                                       # the 512-pixel ping-pong kernel or an error, never a fallback
    t = timed(lambda: ops.conv_igemm(d, x, wb, out), 3)
    kernel = ops.last_igemm_kernel()
    loop_tf = 2.0 * B * H * H * cout * 9 * cin / t / 1e12
    del x, wb, wt, out
    return {"mfma_stream": round(stream_tf, 1), "attainable": round(loop_tf, 1), "attainable_kernel": kernel,
            "attainable_how": "steady-state K loop of the dominant kernel: the same conv3x3_pp_kernel<5,false,0,8> on a "
                              "synthetic 16x32x32 conv, 2560 -> 1280 channels (720 K steps per tile, 32 x 8 = 256 tiles, "
                              "one per CU, operands resident in the Infinity Cache; 0.97 PFLOP per launch), HIP events over 3 launches after the timed loop; "
                              "mfma_stream = register-only 32x32x16 bf16 MFMA stream on 256 CUs x 8 waves "
                              "(nbdt_probe_mfma_stream): the power / clock ceiling"}


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: become the launcher (one rank per GPU, RCCL)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def agreement(eng, num_classes, n, dev):
    """Eval-mode forward of the SAME weights and running statistics on the same `n` synthetic images through
    the engine (bf16 storage / fp32 accumulate) and through the fp32 CPU oracle port; HardNBDT predictions from
    each path's own logits (HIP kernel vs numpy oracle)."""
    nbdt_path.add(oracle=True)
    import numpy as np
    import nbdt_oracle as O
    import torch_models as TM
    from nbdt import _C
    from nbdt.tree import Tree
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(n, 3, 32, 32, generator=g)
    ref = TM.WRN(num_classes, 28, 10)
    ref.load_state_dict({k: v.cpu() for k, v in eng.state_dict().items()})
    ref.eval()
    with torch.no_grad():
        z_ref = torch.cat([ref(x[i:i + 64]) for i in range(0, n, 64)]).numpy()
    z = eng.forward(x.to(dev), training=False).float().cpu().numpy()
    tree = Tree("CIFAR10", hierarchy="induced-wrn28_10_cifar10")
    otree = O.OracleTree(*O.default_paths("CIFAR10", "induced-wrn28_10_cifar10",
                                          os.path.join(nbdt_path.PKG_DIR, "nbdt")))
    hard = _C.hard_forward(tree.device_handle(dev.index), torch.from_numpy(z).to(dev), want_onehot=False)[0]
    hard_ref = O.hard_forward(otree, z_ref)
    scale = float(np.abs(z_ref).max())
    err = float(np.abs(z - z_ref).max()) / scale
    return {"argmax": round(float((z.argmax(1) == z_ref.argmax(1)).mean()), 4),
            "hard_pred": round(float((hard.cpu().numpy() == hard_ref).mean()), 4), "n": n,
            "max_abs_logit_err_over_scale": round(err, 5), "tolerance": LOGIT_TOLERANCE,
            "within_tolerance": bool(err < LOGIT_TOLERANCE),
            "vs": "fp32 torch-CPU port of WRN-28-10 (oracle/torch_models.py) with the engine's trained weights and "
                  "running statistics, eval mode; the reference itself needs pytorchcv (absent) for this backbone"}


def cpu_baseline(batch, num_classes, steps=3):
    """fp32 CPU oracle port timed on this box's host cores: one warm-up step at batch 8, then `steps` timed
    steps at `batch` images (bounded sample of the same workload)."""
    nbdt_path.add(oracle=True)
    import nbdt_oracle as O
    import torch_models as TM
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    net = TM.WRN(num_classes, 28, 10)
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4)
    otree = O.OracleTree(*O.default_paths("CIFAR10", "induced-wrn28_10_cifar10",
                                          os.path.join(nbdt_path.PKG_DIR, "nbdt")))

    def step(b):
        x = torch.randn(b, 3, 32, 32)
        y = torch.randint(0, num_classes, (b,))
        opt.zero_grad()
        z = net(x)
        _, dz = O.soft_tree_sup_loss(otree, z.detach().numpy(), y.numpy())
        z.backward(torch.from_numpy(dz))
        opt.step()

    step(8)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(batch)
    dt = time.perf_counter() - t0
    return {"value": round(steps * batch / dt, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"{steps} training steps of fp32 torch-CPU WRN-28-10 + oracle SoftTreeSupLoss at batch {batch} "
                      f"({dt:.1f} s), after a batch-8 warm-up; a port because the reference's WRN lives in "
                      "pytorchcv and /root/reference does not exist on the GPU box"}


# algorithmic GFLOP per image of a training step (forward 2*MAC of the REAL channel counts, x3 for the two gradients);
# ResNet18 (reference nbdt/models/resnet.py:115-149, 3x3 stem, no max-pool): 555.4 MMAC at 32x32, x4 at 64x64
GFLOP_RESNET18_32 = 3 * 2 * 0.5554
GFLOP_RESNET18_64 = 4 * GFLOP_RESNET18_32


def other_configs(dev, budget_s=0.6):
    """The other BASELINE.json configurations on this GPU, each a <= ~1 s measurement of nbdt.engine.train_step (or the
    eval-mode forward + HardNBDT rules) on one resident synthetic batch: C1 ResNet18/CIFAR10 B=128, C3's per-GPU shard
    (WRN-28-10/CIFAR100, 1024 images over 4 GPUs = 256), C4 ResNet18/TinyImagenet200 64x64 B=128 (SoftTreeSupLoss with
    tree-supervision weight 10, and HardNBDT inference), C5 EfficientNet-B0/Imagenet1000 224x224 B=128.  `frac` is
    against the bound named: bf16 MFMA peak for the ResNets (algorithmic flops of the real channel counts), HBM peak
    for EfficientNet-B0 with the ALGORITHMIC bytes of its schedule (every pass reads each of its input tensors once and
    writes each output once: EfficientNetEngine.algorithmic_bytes -- round 4 quoted 2 x the resident buffers, which left
    out backward's re-reads and was 2.85x below what the counters saw).  `profile`: the committed
    rocprofv3 --kernel-trace --stats summary of the same configuration (scratch/run_config.py <name> --one-stream)."""
    from nbdt import engine as E
    from nbdt.engine_effnet import EfficientNetEngine
    from nbdt.loss import SoftTreeSupLoss
    from nbdt.model import HardEmbeddedDecisionRules
    from nbdt.tree import Tree

    def timeit(fn):
        # The small configurations run close to the host's launch rate (ResNet18 / CIFAR10: 1.6 ms of enqueue per 2.3 ms step),
        # so a generation-2 collection of Python's cyclic GC inside the <= 0.1 s window shows: one run in four read 3.5 ms
        # instead of 2.3 (scratch/c1_after_wrn.py: the next 40 steps of the same engine read 2.3 again).  Collect first and keep
        # the collector off for the window, as the standard library's timeit does.
        import gc
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        one = time.perf_counter() - t0
        steps = max(3, min(40, int(budget_s / max(one, 1e-4))))
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        finally:
            if was_enabled:
                gc.enable()
        return dt, steps

    def entry(name, B, dt, steps, gflop_img=None, bytes_step=None, profile=None, mode="train step"):
        e = {"config": name, "mode": mode, "batch_per_gpu": B, "value": round(B / dt, 1), "unit": "images/sec",
             "ms_per_step": round(1e3 * dt, 3), "steps": steps}
        if gflop_img is not None:
            tf = B / dt * gflop_img / 1e3
            e.update({"bound": "mfma", "gflop_per_image": round(gflop_img, 3), "achieved": round(tf, 1),
                      "peak": PEAK_BF16_TFLOPS, "peak_unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4)})
        if bytes_step is not None:
            gbs = bytes_step / dt / 1e9
            e.update({"bound": "hbm", "bytes_per_step_algorithmic": int(bytes_step), "achieved": round(gbs, 1),
                      "peak": 8000.0, "peak_unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                      "bytes_model": "every pass of the schedule reads each input tensor once and writes each output "
                                     "once (EfficientNetEngine.algorithmic_bytes: activations re-read by backward, "
                                     "two-tensor passes counted as such); round 4 quoted 2 x the resident buffers"})
        if profile:
            e["profile"] = profile
        return e

    def train_case(name, eng, dataset, hierarchy, B, size, C, tsw, **kw):
        crit = SoftTreeSupLoss(dataset=dataset, criterion=nn.CrossEntropyLoss(), hierarchy=hierarchy,
                               tree_supervision_weight=tsw)
        g = torch.Generator().manual_seed(7)
        x = torch.randn(B, 3, size, size, generator=g).to(dev)
        y = torch.randint(0, C, (B,), generator=g).to(dev)
        dt, steps = timeit(lambda: E.train_step(eng, crit, x, y, 0.01))
        if kw.pop("algorithmic_bytes", False):
            kw["bytes_step"] = eng.algorithmic_bytes(B, size)
        return entry(name, B, dt, steps, **kw)

    out = []
    out.append(train_case("C1 ResNet18 + SoftTreeSupLoss, CIFAR10 (10 leaves)", E.ResNetEngine(10, device=dev),
                          "CIFAR10", "induced-ResNet18", 128, 32, 10, 1.0, gflop_img=GFLOP_RESNET18_32,
                          profile="profiles/r06_c1_kernel_stats_one_stream.txt"))
    out.append(train_case("C3 WideResNet28x10 + SoftTreeSupLoss, CIFAR100 (100 leaves): one GPU's 256-image share of "
                          "batch 1024 on 4 GPUs", E.WRNEngine(100, device=dev), "CIFAR100",
                          "induced-wrn28_10_cifar100", 256, 32, 100, 1.0, gflop_img=GFLOP_PER_IMG_TRAIN,
                          profile="profiles/r06_c3_kernel_stats_one_stream.txt"))
    eng = E.ResNetEngine(200, device=dev)
    out.append(train_case("C4 ResNet18 + SoftTreeSupLoss (tree-supervision weight 10), TinyImagenet200 64x64 (200 "
                          "leaves)", eng, "TinyImagenet200", "induced-ResNet18", 128, 64, 200, 10.0,
                          gflop_img=GFLOP_RESNET18_64, profile="profiles/r06_c4_kernel_stats_one_stream.txt"))
    out.append(train_case("C4 ResNet18 + SoftTreeSupLoss (tree-supervision weight 10), TinyImagenet200 64x64, 512 images "
                          "per GPU (BASELINE.json leaves the batch open; 128 x 64x64 gives the 256 CUs 32-256 tiles per launch)",
                          eng, "TinyImagenet200", "induced-ResNet18", 512, 64, 200, 10.0,
                          gflop_img=GFLOP_RESNET18_64, profile="profiles/r06_c4_kernel_stats_one_stream.txt"))
    rules = HardEmbeddedDecisionRules(tree=Tree("TinyImagenet200", hierarchy="induced-ResNet18"))
    x = torch.randn(128, 3, 64, 64, device=dev)
    dt, steps = timeit(lambda: rules.predict(eng.forward(x, training=False)))
    out.append(entry("C4 ResNet18 + HardNBDT (argmax path), TinyImagenet200 64x64", 128, dt, steps,
                     gflop_img=GFLOP_RESNET18_64 / 3, profile="profiles/r06_c4inf_kernel_stats_one_stream.txt",
                     mode="inference: eval-mode backbone + hard decision rules"))
    del eng
    eng = EfficientNetEngine(1000, device=dev)
    out.append(train_case("C5 EfficientNet-B0 + SoftTreeSupLoss, Imagenet1000 induced hierarchy (1000 leaves), 224x224",
                          eng, "Imagenet1000", "induced-efficientnet_b7b", 128, 224,
                          1000, 1.0, algorithmic_bytes=True, profile="profiles/r06_c5_kernel_stats_one_stream.txt"))
    out.append(train_case("C5 EfficientNet-B0 + SoftTreeSupLoss, Imagenet1000 induced hierarchy (1000 leaves), 224x224, 512 "
                          "images per GPU (BASELINE.json leaves the batch open; at 128 images the 16 units' ~470 launches "
                          "average 21 us and ~160 of them are 4-5 us folds)",
                          eng, "Imagenet1000", "induced-efficientnet_b7b", 512, 224,
                          1000, 1.0, algorithmic_bytes=True, profile="profiles/r06_c5_kernel_stats_one_stream.txt"))
    del eng
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="images per GPU (weak scaling)")
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="every launch on one stream (profiling passes: a kernel's duration is then its own)")
    ap.add_argument("--no-cu-share", action="store_true",
                    help="weight gradients beside the data gradients, BatchNorm-backward passes on all CUs "
                         "(engine.set_cu_share(None)); A/B of the CU-sharing schedule the engine runs by default")
    ap.add_argument("--cu-share-force", action="store_true",
                    help="CU sharing without the calibration (counter-collection passes serialise the kernels, so a "
                         "calibration under rocprofv3 --pmc would turn the schedule off)")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="torch.distributed backend for --gpus > 1 (default: nccl = RCCL).  gloo accepts device "
                         "tensors, so `--gpus 2 --backend gloo --share-gpu` runs the N-rank branch on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="every rank on cuda:0 (tests on a 1-GPU box)")
    ap.add_argument("--agreement-n", type=int, default=512, help="images in the prediction-agreement check (0: skip)")
    ap.add_argument("--no-attainable", action="store_true",
                    help="skip the two probes behind roofline.attainable / roofline.mfma_stream (~0.1 s of GPU time)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the <= 1 s measurements of the other BASELINE.json configurations (`other_configs`)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    from nbdt import dist as ndist
    from nbdt import engine as E
    from nbdt import ops
    from nbdt.loss import SoftTreeSupLoss

    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
    if args.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = ndist.init_from_env(args.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    comm = ndist.GradComm() if world > 1 else None

    eng = E.WRNEngine(num_classes=args.classes, blocks=28, width_factor=10, device=dev, seed=0)
    crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(),
                           hierarchy="induced-wrn28_10_cifar10")
    if args.no_overlap:
        eng.set_overlap(False)
    # BatchNorm-backward passes on ~100 CUs beside the weight gradients on the other ~160 is the engine's default
    # (engine.set_cu_share): before the first backward of the warm-up the engine times one conv's backward in both
    # orders and keeps the schedule only if it is faster on this box; with N ranks, rank 0 decides for all
    if args.no_cu_share:
        eng.set_cu_share(None)
    elif args.cu_share_force:
        eng.set_cu_share(47.0, calibrate=False)
    g = torch.Generator().manual_seed(1234 + rank)
    img = torch.randn(args.batch, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, args.classes, (args.batch,), generator=g).to(dev)
    lr = 0.01

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        E.train_step(eng, crit, img, y, lr, comm=comm)
    import gc
    gc.collect()      # (a generation-2 collection is tens of ms of host time: start the timed steps with an empty backlog)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = E.train_step(eng, crit, img, y, lr, comm=comm)
    sync()
    dt = time.perf_counter() - t0
    loss_val = loss.item()

    # Per-kernel durations for the roofline object: the SAME step loop continues for a few more steps
    # with a HIP event pair around every conv_igemm / conv_wgrad launch on the launch stream.  They are
    # kept out of the K steps that define `value` because 174 event records per step sit between
    # back-to-back kernels and cost ~1.2 ms/step (measured: 21.1 ms -> 22.3 ms), a 5 % bias.
    timer = None if args.no_kernel_timer else ops.KernelTimer()
    roof_steps = min(args.steps, 5)
    if timer is not None:
        # The engine normally runs weight gradients on a second stream next to the data-gradient chain; two
        # kernels sharing the GPU stretch each other's wall time, so for a launch's duration to be ITS OWN the
        # roofline pass puts every launch back on one stream (this costs ~3 % of step time, not counted anywhere).
        eng.set_overlap(False)
        ops.set_timer(timer)
        # ... and the rules layer: HIP events around the fused head launch (classifier + rules + SoftTreeSupLoss
        # forward and backward, nbdt_head_soft_tree_loss) of the same steps
        head_events = []
        real_head = crit.head_loss_and_grad

        def timed_head(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = real_head(*a, **k)
            e1.record()
            head_events.append((e0, e1))
            return r

        crit.head_loss_and_grad = timed_head
        try:
            for _ in range(roof_steps):
                E.train_step(eng, crit, img, y, lr, comm=comm)
            sync()
        finally:
            crit.head_loss_and_grad = real_head
        ops.set_timer(None)
        eng.set_overlap(not args.no_overlap)

    # The weight gradients as they run IN the timed step: on the second stream, sized for the CUs the confined
    # BatchNorm passes leave them.  Events are recorded on the stream the launch goes to (ops.conv_wgrad).
    in_step = None
    if timer is not None and not args.no_overlap:
        in_step = ops.KernelTimer(only=("conv_wgrad",))
        ops.set_timer(in_step)
        for _ in range(roof_steps):
            E.train_step(eng, crit, img, y, lr, comm=comm)
        sync()
        ops.set_timer(None)

    dt_nocomm = None
    if world > 1:
        # replicas must still be bit-identical here: same initial weights, every rank applied the SAME all-reduced
        # gradient in every step so far (timed loop and roofline passes alike); two checksums per rank, gathered
        flat = eng.store.flat.double()
        sums = [None] * world
        torch.distributed.all_gather_object(sums, (float(flat.sum().item()), float((flat * flat).sum().item())))
        del flat
        # what the gradient exchange costs a step: the same loop without it (ranks drift apart, nothing after
        # this point reads the weights except the per-rank roofline pass above, which already ran)
        nc_steps = min(args.steps, 5)
        sync()
        t1 = time.perf_counter()
        for _ in range(nc_steps):
            E.train_step(eng, crit, img, y, lr, comm=None)
        sync()
        dt_nocomm = (time.perf_counter() - t1) / nc_steps
        t = torch.tensor([dt, dt_nocomm], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, dt_nocomm = t.tolist()
        shares = [None] * world          # every rank's schedule (rank 0's calibration decides for all)
        torch.distributed.all_gather_object(shares, bool(eng._cu_share is not None))
    if rank != 0:
        return

    ms = 1e3 * dt / args.steps
    value = args.batch * world * args.steps / dt
    out = {
        "metric": "images/sec training SoftNBDT WRN28x10 CIFAR10",
        "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "WideResNet-28-10 + SoftTreeSupLoss (induced-wrn28_10_cifar10, 10 leaves), "
                               "CIFAR10-shaped 3x32x32, SGD m=0.9 wd=5e-4",
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                   "parallelism": f"dp{world}", "final_loss": round(loss_val, 4)},
        "step_mfma_frac": round(value / world * GFLOP_PER_IMG_TRAIN / 1e3 / PEAK_BF16_TFLOPS, 4),
    }
    if eng.cu_share_report is not None:
        out["cu_share"] = eng.cu_share_report
    if world > 1:
        out["comm"] = {"backend": torch.distributed.get_backend(), "ranks": torch.distributed.get_world_size(),
                       "cu_share_per_rank": shares,
                       "replica_checksums": sums, "replicas_identical": all(c == sums[0] for c in sums),
                       "allreduce_bytes_per_rank": int(eng.store.grad.numel()) * 4, "buckets": 3,
                       "ms_per_step_without_allreduce": round(1e3 * dt_nocomm, 3),
                       "allreduce_ms_exposed": round(ms - 1e3 * dt_nocomm, 3)}
        out["comm"].update(comm.describe(int(eng.store.grad.numel()) * 4))
        out["comm"]["grad_buckets_planned"] = [list(r) for r in eng.grad_buckets()]      # [stage 3 .. head], [stage 2], [stem .. stage 1]
    if timer is not None:
        summ = timer.summary()
        traffic, traffic_src = pmc_traffic({k: v / roof_steps for k, v in timer.kernels.items()})
        k = summ.get("conv_igemm")
        if k:
            out["roofline"] = {"bound": "mfma",
                               "kernel": "conv_igemm: conv3x3_pp_kernel (41 dense 3x3 launches) / conv_seg_kernel (the "
                                         "shape-changing units: strided conv1 over the space-to-depth input, conv2 + "
                                         "shortcut in one launch, conv1 + shortcut data gradients in one launch; 8 launches) "
                                         "/ conv_igemm_dma_kernel (one 1x1) -- every forward + data-gradient implicit GEMM, "
                                         "2/3 of the step's flops",
                               "achieved": round(k["tflops"], 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(k["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                               "traffic_source": traffic_src,
                               "avg_launch_us": round(k["avg_us"], 1),
                               "launches_per_step": k["launches"] // roof_steps,
                               "flops_per_launch_avg": k["flops"] / k["launches"],
                               "measured_over": f"{roof_steps} steps continuing the timed loop, HIP events "
                                                "around every launch on the launch stream, weight-gradient "
                                                "stream joined (no concurrent kernel)"}
            if world == 1 and not args.no_attainable:
                try:
                    a = attainable(dev)
                    out["roofline"].update(a)
                    out["roofline"]["frac_of_attainable"] = round(k["tflops"] / a["attainable"], 4)
                    out["roofline"]["attainable_frac_of_peak"] = round(a["attainable"] / PEAK_BF16_TFLOPS, 4)
                    out["roofline"]["mfma_stream_frac_of_peak"] = round(a["mfma_stream"] / PEAK_BF16_TFLOPS, 4)
                    out["roofline"]["frac_of_mfma_stream"] = round(k["tflops"] / a["mfma_stream"], 4)
                except Exception as exc:          # a probe must never cost the bench line
                    out["roofline"]["attainable"] = None
                    out["roofline"]["attainable_error"] = repr(exc)[:300]
        w = summ.get("conv_wgrad")
        if w:
            out["roofline_wgrad"] = {"bound": "mfma",
                                     "kernel": "conv_wgrad: conv_wgrad_ks_kernel (23 of 28 launches) / conv_wgrad_s2d_kernel "
                                               "(the two strided 3x3) / conv_wgrad_dma_kernel (the three 1x1 shortcuts)",
                                     "achieved": round(w["tflops"], 1),
                                     "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(w["tflops"] / PEAK_BF16_TFLOPS, 4),
                                     "avg_launch_us": round(w["avg_us"], 1),
                                     "launches_per_step": w["launches"] // roof_steps,
                                     "measured": "alone: one stream, all 256 CUs (cu_budget 0)"}
            wi = in_step.summary().get("conv_wgrad") if in_step is not None else None
            if wi:   # the launch the timed step makes: second stream, CU-budgeted, a confined BatchNorm pass beside it
                out["roofline_wgrad"].update({
                    "in_step_us": round(wi["avg_us"], 1), "in_step_achieved": round(wi["tflops"], 1),
                    "in_step_frac": round(wi["tflops"] / PEAK_BF16_TFLOPS, 4),
                    "in_step_measured": "HIP events on the second stream around every weight-gradient launch of "
                                        f"{roof_steps} further steps of the timed schedule (CU-budgeted launches beside "
                                        "the CU-confined BatchNorm passes; fraction of the FULL chip's peak)"})
            out["step_ms_in_mfma_kernels"] = round((k["ms"] + w["ms"]) / roof_steps, 3) if k else None
        if head_events:
            # Rules layer computed from features (north star): algorithmic bytes 4 (B F + R F + B C) and flops
            # 2 B R F with R = child slots of the hierarchy -- the survey's formula (the backward half's dL/dpooled write and
            # classifier-gradient atomics are of the same order and not counted).  The honest bound is one kernel boundary.
            us = 1e3 * sum(a.elapsed_time(b) for a, b in head_events) / len(head_events)
            Bh, F_, C_ = args.batch, eng.feat_c, args.classes
            R_ = int(crit.tree.flat.num_slots)
            nbytes = 4 * (Bh * F_ + R_ * F_ + Bh * C_)
            out["roofline_rules"] = {"bound": "hbm", "kernel": "head_soft_loss_kernel (nbdt_head_soft_tree_loss): classifier "
                                     "forward + node logits + per-node softmax / path products + SoftTreeSupLoss + the "
                                     "classifier's backward, one launch; the logits never reach HBM",
                                     "us_per_step": round(us, 1), "algorithmic_bytes": nbytes,
                                     "achieved": round(nbytes / (us * 1e-6) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                                     "frac": round(nbytes / (us * 1e-6) / 8e12, 6),
                                     "flops": 2 * Bh * R_ * F_, "child_slots": R_,
                                     "stated_bound": "one kernel boundary (~1.5 us): 1.3 MB at 8 TB/s is 0.17 us -- the launch is "
                                                     "latency-bound (512 samples x 18 child slots; 4 samples per block), not "
                                                     "bandwidth-bound",
                                     "share_of_step": round(us * 1e-3 / ms, 5),
                                     "measured": f"HIP events around the launch in {roof_steps} steps continuing the timed loop "
                                                 "(one stream)"}
    if world == 1 and args.agreement_n > 0:
        out["agreement"] = agreement(eng, args.classes, args.agreement_n, dev)
    if world == 1 and not args.no_other_configs:
        del eng      # (its 7 GB of activations; everything that reads it has run)
        out["other_configs"] = other_configs(dev)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.classes)
    if world == 1 and "agreement" in out and not out["agreement"]["within_tolerance"]:
        print(f"bench.py: logit error {out['agreement']['max_abs_logit_err_over_scale']} of the logit scale exceeds the "
              f"stated tolerance {LOGIT_TOLERANCE}", file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
