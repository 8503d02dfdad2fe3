#!/usr/bin/env python3
"""Training / evaluation driver with the reference's ``main.py`` command line.

Mirrors reference main.py:28-90 (flags), :164-188 (``--resume`` / ``--path-resume`` with checkpoints
``{"net", "acc", "epoch"}`` and ``module.`` prefix coercion), :191-210 (loss construction from
``--loss``, SGD momentum 0.9 wd 5e-4, MultiStepLR at 3/7 and 5/7 of ``--epochs``), :218-308 (train /
test loops, best-accuracy checkpointing to ``./checkpoint/<generate_checkpoint_fname>.pth``), so the
recipes in the reference's ``scripts/*.sh`` keep their arguments.  What differs, and why:

* One process per GPU instead of ``DataParallel`` (launch with ``python -m torch.distributed.run
  --nproc-per-node N main.py ...``): each rank trains on its shard, gradients are all-reduced (RCCL).
* The step itself is the engine's fixed launch sequence: forward, fused tree-supervision loss
  (loss + dL/dlogits in one kernel), backward, fused SGD -- no autograd graph, no optimizer object.
* Datasets: torchvision is not available on the target image and the reference's dataset/transform
  layer is out of scope, so samples come from ``--data-file`` (a ``torch.save``d
  dict with ``train_x [N,3,H,W]`` uint8|float, ``train_y``, ``test_x``, ``test_y``; already
  normalised if float, scaled to [0,1] and normalised with the reference's CIFAR statistics if uint8)
  or from ``--synthetic N`` (CIFAR-shaped noise with a learnable class signal, for smoke runs).
* ``--analysis Noop | SoftEmbeddedDecisionRules | HardEmbeddedDecisionRules`` drives an analyzer of ``nbdt.analysis``
  through the reference's hook protocol (reference main.py:212-288, nbdt/analysis.py:81-130): ``epoch_context`` around
  every epoch, ``start_train`` / ``end_train`` around the training pass, ``start_test`` / ``update_batch(logits,
  targets, images)`` per evaluation batch / ``end_test``.  The two rules analyzers report the accuracy of the decision
  rules applied to the backbone's logits (reference nbdt/analysis.py:204-252) next to the backbone's ``--metric``; their
  counters stay on the device, one host transfer per evaluation.  The training pass does not call ``update_batch``: the
  fused step (classifier + rules + loss + their backward in one launch) never materialises the logits.  The
  reference's presentation analyzers are out of scope.
"""
import argparse
import math
import os
import sys
from pathlib import Path

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from nbdt import analysis  # noqa: E402
from nbdt import dist as ndist  # noqa: E402
from nbdt import loss as losses  # noqa: E402
from nbdt import models  # noqa: E402
from nbdt.engine import train_step  # noqa: E402
from nbdt.model import coerce_state_dict  # noqa: E402
from nbdt.tree import Tree  # noqa: E402
from nbdt.utils import DATASET_TO_NUM_CLASSES, DATASETS  # noqa: E402

CIFAR_MEAN, CIFAR_STD = (0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010)   # reference nbdt/data/cifar.py:17-19
METRICS = {"top1": 1, "top2": 2, "top5": 5, "top10": 10}      # the reference's --metric names -> k


class HitCounter:
    """Running count of samples whose target is among the k largest scores.  Both counters stay on the device;
    `percent()` is the only host transfer."""

    def __init__(self, k, device):
        self.k = k
        self.hits = torch.zeros((), dtype=torch.long, device=device)
        self.seen = 0

    def add(self, scores, targets):
        k = min(self.k, scores.shape[1])
        if k == 1:
            self.hits += (scores.argmax(dim=1) == targets).sum()
        else:
            self.hits += (scores.topk(k, dim=1).indices == targets[:, None]).any(dim=1).sum()
        self.seen += targets.shape[0]

    def percent(self):
        return 100.0 * int(self.hits) / max(self.seen, 1)


def build_parser():
    p = argparse.ArgumentParser(description="NBDT training on MI355X (reference main.py command line)")
    p.add_argument("--batch-size", default=512, type=int, help="GLOBAL batch size (split over ranks)")
    p.add_argument("--epochs", "-e", default=200, type=int, help="lr schedule is scaled accordingly")
    p.add_argument("--dataset", default="CIFAR10", choices=DATASETS)
    p.add_argument("--arch", default="ResNet18", choices=models.get_model_choices())
    p.add_argument("--lr", default=0.1, type=float)
    p.add_argument("--resume", "-r", action="store_true")
    p.add_argument("--path-resume", default="")
    p.add_argument("--name", default="")
    p.add_argument("--pretrained", action="store_true")
    p.add_argument("--eval", action="store_true")
    p.add_argument("--loss", choices=losses.names, default=["CrossEntropyLoss"], nargs="+")
    p.add_argument("--metric", choices=sorted(METRICS), default="top1")
    p.add_argument("--analysis", choices=analysis.names, help="analyzer run during every evaluation (nbdt.analysis)")
    # nbdt/tree.py:26-35
    p.add_argument("--hierarchy")
    p.add_argument("--path-graph")
    p.add_argument("--path-wnids")
    losses.add_arguments(p)      # --xent-weight* / --tree-supervision-weight* / --tree-*-epochs (reference nbdt/loss.py:27-80)
    # data source (see module docstring)
    p.add_argument("--data-file", help="torch.save'd dict: train_x, train_y, test_x, test_y")
    p.add_argument("--synthetic", type=int, default=0, help="number of synthetic training samples")
    p.add_argument("--image-size", type=int, default=0, help="synthetic image size (default: dataset's)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--deterministic", action="store_true",
                   help="bit-reproducible training steps, like the reference's CPU path: every cross-block reduction in "
                        "a fixed order instead of fp32 atomics (nbdt_set_deterministic; ResNet / WideResNet backbones)")
    return p


def generate_checkpoint_fname(dataset, arch, path_graph=None, name="", tree_supervision_weight=1,
                              loss=("CrossEntropyLoss",), lr=0.1, tree_supervision_weight_end=None,
                              tree_supervision_weight_power=1, xent_weight=1, xent_weight_end=None,
                              xent_weight_power=1, **_):
    """reference nbdt/utils.py:266-330 (the parts reachable from this driver's flags)."""
    fname = f"ckpt-{dataset}-{arch}"
    if lr != 0.1:
        fname += f"-lr{lr}"
    if name:
        fname += "-" + name
    if path_graph and any("TreeSupLoss" in l for l in loss):   # (SoftTreeLoss does not match: reference quirk)
        fname += "-" + Path(path_graph).stem.replace("graph-", "", 1)
    if len(loss) > 1 or loss[0] != "CrossEntropyLoss":
        fname += f'-{",".join(loss)}'
        if tree_supervision_weight not in (None, 1):
            fname += f"-tsw{tree_supervision_weight}"
        if tree_supervision_weight_end not in (tree_supervision_weight, None):
            fname += f"-tswe{tree_supervision_weight_end}"
        if tree_supervision_weight_power not in (None, 1):
            fname += f"-tswp{tree_supervision_weight_power}"
        if xent_weight not in (None, 1):
            fname += f"-xw{xent_weight}"
        if xent_weight_end not in (xent_weight, None):
            fname += f"-xwe{xent_weight_end}"
        if xent_weight_power not in (None, 1):
            fname += f"-xwp{xent_weight_power}"
    return fname


def multistep_lr(base_lr, epoch, epochs, gamma=0.1):
    """optim.lr_scheduler.MultiStepLR(milestones=[int(3/7*E), int(5/7*E)]) -- reference main.py:208-210."""
    milestones = (int(3 / 7.0 * epochs), int(5 / 7.0 * epochs))
    return base_lr * gamma ** sum(epoch >= m for m in milestones)


def build_criterion(args, tree, net=None, checkpoint_path="./"):
    """reference main.py:191-205: the LAST entry of --loss wraps nn.CrossEntropyLoss()."""
    criterion = nn.CrossEntropyLoss()
    for name in args.loss:
        cls = getattr(losses, name)
        if name == "CrossEntropyLoss":
            criterion = cls()
            continue
        kwargs = {"dataset": args.dataset, "criterion": criterion, "tree": tree}
        for key in ("tree_supervision_weight", "tree_supervision_weight_end", "tree_supervision_weight_power",
                    "xent_weight", "xent_weight_end", "xent_weight_power"):
            if getattr(args, key) is not None:
                kwargs[key] = getattr(args, key)
        if name == "SoftTreeLoss":     # mid-training re-induction needs the network and a directory
            kwargs.update(net=net, arch=args.arch, checkpoint_path=checkpoint_path)
            for key in ("tree_start_epochs", "tree_update_every_epochs", "tree_update_end_epochs"):
                if getattr(args, key) is not None:
                    kwargs[key] = getattr(args, key)
        criterion = cls(**kwargs)
    return criterion


class _PlainCE:
    """--loss CrossEntropyLoss on the engine's fast path: the fused kernel with a zero tree weight."""

    def __init__(self, tree):
        self.inner = losses.SoftTreeSupLoss(dataset=None, criterion=nn.CrossEntropyLoss(), tree=tree,
                                            tree_supervision_weight=0.0)

    def set_epoch(self, cur, total):
        pass

    def loss_and_grad(self, z, y, grad_scale=1.0):
        return self.inner.loss_and_grad(z, y, grad_scale)


def load_data(args, num_classes, device):
    if args.data_file:
        blob = torch.load(args.data_file, map_location="cpu")
        out = []
        for split in ("train", "test"):
            x, y = blob[f"{split}_x"], blob[f"{split}_y"].long()
            if x.dtype == torch.uint8:
                x = x.float().div_(255.0)
                mean = torch.tensor(CIFAR_MEAN).view(1, 3, 1, 1)
                std = torch.tensor(CIFAR_STD).view(1, 3, 1, 1)
                x = (x - mean) / std
            out += [x.float().contiguous(), y]
        return out
    n = args.synthetic or 4 * args.batch_size
    size = args.image_size or (64 if args.dataset == "TinyImagenet200" else 224 if args.dataset == "Imagenet1000" else 32)
    g = torch.Generator().manual_seed(args.seed + 17)
    proto = torch.randn(num_classes, 3, size, size, generator=g)      # one pattern per class + noise

    def make(m):
        y = torch.randint(0, num_classes, (m,), generator=g)
        return (proto[y] + torch.randn(m, 3, size, size, generator=g)).contiguous(), y
    return [*make(n), *make(max(n // 4, args.batch_size))]


def evaluate(net, criterion_module, analyzer, k, x, y, batch, device):
    """reference main.py:262-277: top-k accuracy of the backbone's logits and the mean loss over (x, y); every batch's
    logits also go to the analyzer (update_batch), which keeps its own statistic."""
    net.eval()
    plain = HitCounter(k, device)
    loss_sum = torch.zeros((), device=device)
    batches = 0
    with torch.no_grad():
        for i in range(0, x.shape[0], batch):
            xb, yb = x[i:i + batch].to(device), y[i:i + batch].to(device)
            z = net(xb)
            loss_sum += criterion_module(z, yb)
            batches += 1
            plain.add(z, yb)
            analyzer.update_batch(z, yb, xb)
    return plain.percent(), float(loss_sum) / max(batches, 1)


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, world, local = ndist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("main.py needs an MI355X: the NBDT hot path has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    log = print if rank == 0 else (lambda *a, **k: None)
    if args.pretrained:
        raise SystemExit("--pretrained downloads release checkpoints; load a local file with --resume --path-resume")
    if args.batch_size % world:
        raise SystemExit(f"--batch-size {args.batch_size} must be divisible by the number of ranks ({world})")

    num_classes = DATASET_TO_NUM_CLASSES[args.dataset]
    log("==> Preparing data..")
    train_x, train_y, test_x, test_y = load_data(args, num_classes, device)
    log(f"Training with dataset {args.dataset} and {num_classes} classes: {train_x.shape[0]} train / "
        f"{test_x.shape[0]} test samples of shape {tuple(train_x.shape[1:])}")

    log("==> Building model..")
    if args.deterministic:
        from nbdt import ops
        ops.set_deterministic(True)
    net = getattr(models, args.arch)(num_classes=num_classes, device=device, seed=args.seed)
    engine = net.engine

    if not args.hierarchy and not args.path_graph and any("Tree" in l for l in args.loss) or args.analysis:
        args.hierarchy = args.hierarchy or f"induced-{args.arch}"       # reference nbdt/model.py:296-298 default
    tree = Tree.create_from_args(args)
    ck_args = dict(vars(args))
    ck_args["path_graph"] = tree.path_graph
    checkpoint_fname = generate_checkpoint_fname(**ck_args)
    checkpoint_path = f"./checkpoint/{checkpoint_fname}.pth"
    log(f"==> Checkpoints will be saved to: {checkpoint_path}")

    best_acc, start_epoch = 0.0, 0
    resume_path = args.path_resume or checkpoint_path
    if args.resume:
        log("==> Resuming from checkpoint..")
        if not os.path.exists(resume_path):
            log("==> No checkpoint found. Skipping...")
        else:
            checkpoint = torch.load(resume_path, map_location="cpu")
            state = coerce_state_dict(checkpoint, net.state_dict())     # {"net": ...} and `module.` prefix
            net.load_state_dict(state)
            net._sync_mirrors()          # the engine reads the bf16 mirror / dgrad copies: refresh them now
            if "net" in checkpoint:
                best_acc, start_epoch = checkpoint["acc"], checkpoint["epoch"]
                log(f"==> Checkpoint found for epoch {start_epoch} with accuracy {best_acc} at {resume_path}")
            else:
                log(f"==> Checkpoint found at {resume_path}")

    criterion = build_criterion(args, tree, net=net, checkpoint_path=checkpoint_path)
    fast = criterion if hasattr(criterion, "loss_and_grad") else _PlainCE(tree)
    analyzer_cls = getattr(analysis, args.analysis or "Noop")
    analyzer = analyzer_cls(tree=tree, metric=args.metric) if args.analysis not in (None, "Noop") else analyzer_cls(tree.classes)
    analyzer.verbose = rank == 0                  # one rank prints
    comm = ndist.GradComm() if world > 1 else None
    per_rank = args.batch_size // world

    @analyzer.train_function
    def train(epoch):
        if hasattr(criterion, "set_epoch"):
            criterion.set_epoch(epoch, args.epochs)
        lr = multistep_lr(args.lr, epoch, args.epochs)
        log("\nEpoch: %d / LR: %.04f" % (epoch, lr))
        net.train()
        g = torch.Generator().manual_seed(args.seed * 1000 + epoch)       # same shuffle on every rank
        perm = torch.randperm(train_x.shape[0], generator=g)
        steps = train_x.shape[0] // args.batch_size
        total = torch.zeros((), device=device)
        for i in range(steps):
            idx = perm[i * args.batch_size:(i + 1) * args.batch_size]
            idx = ndist.shard_batch(idx, rank, world)
            xb, yb = train_x[idx].to(device, non_blocking=True), train_y[idx].to(device, non_blocking=True)
            total += train_step(engine, fast, xb, yb, lr, comm=comm)
        log("Loss: %.3f (%d steps of %d x %d images)" % (total.item() / max(steps, 1), steps, world, per_rank))

    def test(epoch, checkpoint=True):
        nonlocal best_acc
        analyzer.start_test(epoch)
        acc, loss = evaluate(net, criterion, analyzer, METRICS[args.metric], test_x, test_y, 100, device)
        nbdt_acc = analyzer.accuracy() if hasattr(analyzer, "accuracy") else None
        extra = f" | {analyzer.name}: {nbdt_acc:.3f}%" if nbdt_acc is not None else ""
        log("Loss: %.3f | Acc: %.3f%%%s" % (loss, acc, extra))
        analyzer.end_test(epoch)
        log(f"Accuracy: {acc} | Best Accuracy: {best_acc}")
        if acc > best_acc and checkpoint and rank == 0:
            log(f"Saving to {checkpoint_fname} ({acc})..")
            os.makedirs("checkpoint", exist_ok=True)
            torch.save({"net": {k: v.cpu() for k, v in net.state_dict().items()}, "acc": acc, "epoch": epoch},
                       checkpoint_path)
        best_acc = max(best_acc, acc)
        return acc, nbdt_acc

    if args.eval:
        if not args.resume:
            log(" * Warning: Model is not loaded from checkpoint. Use --resume")
        with analyzer.epoch_context(0):
            return test(0, checkpoint=False)
    result = None
    for epoch in range(start_epoch, args.epochs):
        with analyzer.epoch_context(epoch):
            train(epoch)
            result = test(epoch)
    return result


if __name__ == "__main__":
    main()
