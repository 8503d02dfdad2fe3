"""Tree-supervision losses, MI355X-native.

Drop-in surface of the reference's ``nbdt/loss.py``: ``TreeSupLoss`` (:97-209) and
``SoftTreeSupLoss`` (:260-266) with the same constructor arguments, ``accepts_*`` class
attributes (used by the reference's ``main.py`` flag plumbing), ``set_epoch`` / ``get_weight``
weight schedule and the ``_nbdt_output_flag`` guard.

When the wrapped criterion is a default ``nn.CrossEntropyLoss()`` the whole loss --
``w_x*CE(z,y) + w_t*CE(rules(z), y)`` and its gradient -- is ONE fused HIP kernel
(csrc/rules.hip: soft_loss_kernel).  Any other criterion composes the fused rules kernel
(autograd-enabled) with the user's criterion, exactly like the reference.

``HardTreeSupLoss`` (reference :212-257) gets the same treatment (hard_loss_kernel).  ``SoftTreeLoss``
(:269-315) adds mid-training re-induction of the hierarchy (nbdt/graph.py, nbdt/hierarchy.py).
"""
import torch
import torch.nn as nn

from nbdt import _C
from nbdt.model import HardEmbeddedDecisionRules, SoftEmbeddedDecisionRules
from nbdt.tree import Tree
from nbdt.utils import dataset_to_default_path_graph, dataset_to_default_path_wnids, hierarchy_to_path_graph

__all__ = names = ("HardTreeSupLoss", "SoftTreeSupLoss", "SoftTreeLoss", "CrossEntropyLoss")

CrossEntropyLoss = nn.CrossEntropyLoss


def add_arguments(parser):
    """The loss flags of the training driver (reference nbdt/loss.py:27-80): schedule end points / powers of the two
    weights, and SoftTreeLoss's hierarchy re-induction epochs."""
    for flags, kw in (
            (("--xent-weight", "--xw"), dict(type=float, help="weight of the cross-entropy term")),
            (("--xent-weight-end", "--xwe"), dict(type=float, help="its value at the end of training (default: constant)")),
            (("--xent-weight-power", "--xwp"), dict(type=float, help="training progress is raised to this power")),
            (("--tree-supervision-weight", "--tsw"), dict(type=float, default=1, help="weight of the tree-supervision term")),
            (("--tree-supervision-weight-end", "--tswe"), dict(type=float, help="its value at the end of training (default: constant)")),
            (("--tree-supervision-weight-power", "--tswp"), dict(type=float, help="> 1 approaches the end value later, < 1 sooner")),
            (("--tree-start-epochs", "--tse"), dict(type=int, help="SoftTreeLoss: first epoch with a tree term (the hierarchy is induced there)")),
            (("--tree-update-end-epochs", "--tuene"), dict(type=int, help="SoftTreeLoss: last epoch that re-induces the hierarchy")),
            (("--tree-update-every-epochs", "--tueve"), dict(type=int, help="SoftTreeLoss: re-induce the hierarchy every this many epochs"))):
        parser.add_argument(*flags, **kw)


def set_default_values(args):
    """Resolve --hierarchy / --path-graph / --path-wnids to files (reference nbdt/loss.py:83-91): a named hierarchy and an
    explicit graph file exclude each other; without either the dataset's default induced hierarchy is used."""
    if getattr(args, "hierarchy", None) and getattr(args, "path_graph", None):
        raise AssertionError("Only one, between --hierarchy and --path-graph can be provided.")
    if getattr(args, "hierarchy", None):
        args.path_graph = hierarchy_to_path_graph(args.dataset, args.hierarchy)
    if not getattr(args, "path_graph", None):
        args.path_graph = dataset_to_default_path_graph(args.dataset)
    if not getattr(args, "path_wnids", None):
        args.path_wnids = dataset_to_default_path_wnids(args.dataset)


def _is_plain_cross_entropy(criterion):
    return (type(criterion) is nn.CrossEntropyLoss and criterion.weight is None
            and criterion.reduction == "mean" and criterion.ignore_index == -100
            and getattr(criterion, "label_smoothing", 0.0) == 0.0)


def _is_class_index(targets):
    """Soft / probability targets ([B, C] floats) take the composed path, as with the reference's criterion."""
    return targets.dim() == 1 and not targets.is_floating_point()


class _FusedSoftTreeLossFn(torch.autograd.Function):
    """loss, and dloss/dz computed in the same launch (saved for backward)."""

    @staticmethod
    def forward(ctx, z, y, tree, w_xent, w_tree):
        handle = tree.device_handle(z.device.index)
        loss, gz = _C.soft_tree_loss(handle, z, y, w_xent, w_tree)
        ctx.save_for_backward(gz)
        ctx.z_dtype = z.dtype
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (gz,) = ctx.saved_tensors
        return (gz * gloss).to(ctx.z_dtype), None, None, None, None


class TreeSupLoss(nn.Module):
    """reference nbdt/loss.py:97-209."""

    accepts_tree = lambda tree, **kwargs: tree
    accepts_criterion = lambda criterion, **kwargs: criterion
    accepts_dataset = lambda trainset, **kwargs: trainset.__class__.__name__
    accepts_path_graph = True
    accepts_path_wnids = True
    accepts_tree_supervision_weight = True
    accepts_classes = lambda trainset, **kwargs: trainset.classes
    accepts_hierarchy = True
    accepts_tree_supervision_weight_end = True
    accepts_tree_supervision_weight_power = True
    accepts_xent_weight = True
    accepts_xent_weight_end = True
    accepts_xent_weight_power = True

    def __init__(self, dataset, criterion, path_graph=None, path_wnids=None, classes=None,
                 hierarchy=None, Rules=HardEmbeddedDecisionRules, tree=None,
                 tree_supervision_weight=1.0, tree_supervision_weight_end=None,
                 tree_supervision_weight_power=1, xent_weight=1, xent_weight_end=None,
                 xent_weight_power=1):
        super().__init__()
        self.tree = tree or Tree(dataset, path_graph, path_wnids, classes, hierarchy=hierarchy)
        self.num_classes = len(self.tree.classes)
        self.rules = Rules(tree=self.tree)
        self.criterion = criterion
        # two weight schedules, each (value at epoch 0, value at the last epoch, exponent of the progress); the public
        # attribute names are the reference's (a driver may read or set them between epochs)
        for prefix, first, last, power in (("tree_supervision_weight", tree_supervision_weight,
                                            tree_supervision_weight_end, tree_supervision_weight_power),
                                           ("xent_weight", xent_weight, xent_weight_end, xent_weight_power)):
            setattr(self, prefix, first)
            setattr(self, prefix + "_end", first if last is None else last)
            setattr(self, prefix + "_power", power)
        self.epochs, self.progress = 0, 1       # progress 1: the *_end weights, until set_epoch() starts a schedule

    @staticmethod
    def assert_output_not_nbdt(outputs):
        if getattr(outputs, "_nbdt_output_flag", False):
            raise AssertionError(
                "these logits came out of an NBDT wrapper (SoftNBDT / HardNBDT): a tree-supervision loss is computed on "
                "the plain backbone's outputs -- the NBDT wrappers are for validation and inference only")

    def forward_tree(self, outputs, targets):
        raise NotImplementedError()

    def get_weight(self, start, end, power=1):
        """start -> end along progress ** power (reference nbdt/loss.py:191-193)."""
        t = self.progress ** power
        return start + (end - start) * t if t not in (0, 1) else (end if t == 1 else start)

    def current_weights(self):
        """(cross-entropy weight, tree-supervision weight) at the current epoch progress -- the only place the two
        schedules are evaluated (forward(), the fused loss kernels and GraphedStep all ask here)."""
        return (self.get_weight(self.xent_weight, self.xent_weight_end, self.xent_weight_power),
                self.get_weight(self.tree_supervision_weight, self.tree_supervision_weight_end,
                                self.tree_supervision_weight_power))

    def forward(self, outputs, targets):
        w_xent, w_tree = self.current_weights()
        return self.criterion(outputs, targets) * w_xent + self.forward_tree(outputs, targets) * w_tree

    def set_epoch(self, cur, total):
        self.epochs, self.progress = cur, cur / total


class SoftTreeSupLoss(TreeSupLoss):
    """reference nbdt/loss.py:260-266."""

    def __init__(self, *args, Rules=None, **kwargs):
        super().__init__(*args, Rules=SoftEmbeddedDecisionRules, **kwargs)

    def forward_tree(self, outputs, targets):
        self.assert_output_not_nbdt(outputs)
        return self.criterion(self.rules(outputs), targets)

    def forward(self, outputs, targets):
        if _is_plain_cross_entropy(self.criterion) and outputs.dim() == 2 and _is_class_index(targets):
            self.assert_output_not_nbdt(outputs)
            _C.require_gpu(outputs, "SoftTreeSupLoss")
            xent_weight, tree_weight = self.current_weights()
            return _FusedSoftTreeLossFn.apply(outputs, targets, self.tree, float(xent_weight),
                                              float(tree_weight))
        return super().forward(outputs, targets)

    def loss_and_grad(self, outputs, targets, grad_scale=1.0):
        """Engine fast path: (loss, dloss/dz * grad_scale) from one launch, no autograd."""
        self.assert_output_not_nbdt(outputs)
        xent_weight, tree_weight = self.current_weights()
        handle = self.tree.device_handle(outputs.device.index)
        return _C.soft_tree_loss(handle, outputs, targets, float(xent_weight), float(tree_weight),
                                 grad_scale)


    def can_fuse_head(self, num_classes):
        """True when head_loss_and_grad() applies: plain cross entropy, a single-path hierarchy of at most 512 classes
        and 512 child slots (the fused kernel's group of lanes per sample), matching the classifier's width."""
        flat = self.tree.flat
        return (_is_plain_cross_entropy(self.criterion) and flat.num_classes == num_classes
                and flat.num_classes <= 512 and flat.num_slots <= 512 and flat.multi_path_node is None)

    def head_loss_and_grad(self, pooled, weight, bias, targets, grad_weight=None, grad_bias=None, grad_scale=1.0,
                           want_logits=False):
        """Engine fast path with the classifier folded in (nbdt_head_soft_tree_loss, one launch): from the pooled
        features [B, K] and the classifier's weight [C, K] / bias [C], returns (loss, dloss/dpooled * grad_scale,
        logits or None) and ACCUMULATES the classifier's gradients into grad_weight / grad_bias.  Replaces
        linear forward -> loss_and_grad -> linear backward; the logits stay on chip."""
        xent_weight, tree_weight = self.current_weights()
        handle = self.tree.device_handle(pooled.device.index)
        return _C.head_soft_tree_loss(handle, pooled, weight, bias, targets, float(xent_weight), float(tree_weight),
                                      grad_scale, gW=grad_weight, gb=grad_bias, want_logits=want_logits)


class _FusedHardTreeLossFn(torch.autograd.Function):
    """HardTreeSupLoss and dloss/dz from one launch (csrc/rules.hip: hard_loss_kernel)."""

    @staticmethod
    def forward(ctx, z, y, tree, w_xent, w_node):
        handle = tree.device_handle(z.device.index)
        loss, gz = _C.hard_tree_loss(handle, z, y, w_xent, w_node)
        ctx.save_for_backward(gz)
        ctx.z_dtype = z.dtype
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (gz,) = ctx.saved_tensors
        return (gz * gloss).to(ctx.z_dtype), None, None, None, None


class HardTreeSupLoss(TreeSupLoss):
    """reference nbdt/loss.py:212-257: cross entropy at every inner node on the label's path.

    With a default ``nn.CrossEntropyLoss()`` the whole loss and its gradient are one fused kernel.
    Any other criterion follows the reference's structure -- rows pooled by the node's child count,
    one criterion call per pool, weighted ``len(pool)/(B*N/2) * tree_supervision_weight`` -- on top
    of ONE node-logit launch (differentiable) instead of a launch cascade per node.  Like the
    reference, ``tree_supervision_weight`` ends up applied twice (inside ``forward_tree`` and
    again by ``TreeSupLoss.forward``).
    """

    def _node_weight(self, tree_weight):
        return float(tree_weight) * float(self.tree_supervision_weight) * 2.0 / len(self.tree.inodes)

    def forward_tree(self, outputs, targets):
        from nbdt.model import _NodeLogitsFn
        self.assert_output_not_nbdt(outputs)
        _C.require_gpu(outputs, "HardTreeSupLoss")
        num_losses = outputs.size(0) * len(self.tree.inodes) / 2.0
        logits = _NodeLogitsFn.apply(outputs, self.tree)            # [B, R], every node at once
        node_off = self.tree.flat.node_off
        targets_ints = [int(t) for t in targets.cpu().long()]
        pools = {}                                                   # K -> (rows, first slot, target)
        for n, node in enumerate(self.tree.inodes):
            rows, base, tgt = pools.setdefault(node.num_classes, ([], [], []))
            for b, t in enumerate(targets_ints):
                child = node.class_index_to_child_index.get(t)
                if child:
                    rows.append(b)
                    base.append(int(node_off[n]))
                    tgt.append(child[0])
        loss = 0
        dev = outputs.device
        for K, (rows, base, tgt) in pools.items():
            if not rows:
                continue
            rows_t = torch.tensor(rows, device=dev).unsqueeze(1)
            cols_t = torch.tensor(base, device=dev).unsqueeze(1) + torch.arange(K, device=dev)
            outputs_sub = logits[rows_t, cols_t]
            targets_sub = torch.tensor(tgt, device=dev, dtype=torch.long)
            fraction = outputs_sub.size(0) / float(num_losses) * self.tree_supervision_weight
            loss = loss + self.criterion(outputs_sub, targets_sub) * fraction
        return loss

    def forward(self, outputs, targets):
        if _is_plain_cross_entropy(self.criterion) and outputs.dim() == 2 and _is_class_index(targets):
            self.assert_output_not_nbdt(outputs)
            _C.require_gpu(outputs, "HardTreeSupLoss")
            xent_weight, tree_weight = self.current_weights()
            return _FusedHardTreeLossFn.apply(outputs, targets, self.tree, float(xent_weight),
                                              self._node_weight(tree_weight))
        return super().forward(outputs, targets)

    def loss_and_grad(self, outputs, targets, grad_scale=1.0):
        """Engine fast path: (loss, dloss/dz * grad_scale) from one launch, no autograd."""
        self.assert_output_not_nbdt(outputs)
        xent_weight, tree_weight = self.current_weights()
        handle = self.tree.device_handle(outputs.device.index)
        return _C.hard_tree_loss(handle, outputs, targets, float(xent_weight),
                                 self._node_weight(tree_weight), grad_scale)


class SoftTreeLoss(SoftTreeSupLoss):
    """reference nbdt/loss.py:269-315: plain cross entropy until `tree_start_epochs`, then soft tree
    supervision on a hierarchy RE-INDUCED from the network's own classifier weights every
    `tree_update_every_epochs` epochs (until `tree_update_end_epochs`).  Re-induction is host-side ward
    clustering (nbdt/graph.py); the kernels pick the new hierarchy up through a fresh tree handle."""

    accepts_tree_start_epochs = True
    accepts_tree_update_every_epochs = True
    accepts_tree_update_end_epochs = True
    accepts_arch = True
    accepts_net = lambda net, **kwargs: net
    accepts_checkpoint_path = lambda checkpoint_path, **kwargs: checkpoint_path

    def __init__(self, *args, arch=None, checkpoint_path="./", net=None, tree_start_epochs=67,
                 tree_update_every_epochs=10, tree_update_end_epochs=120, **kwargs):
        super().__init__(*args, **kwargs)
        self.start_epochs = tree_start_epochs
        self.update_every_epochs = tree_update_every_epochs
        self.update_end_epochs = tree_update_end_epochs
        self.net = net
        self.arch = arch
        self.checkpoint_path = checkpoint_path

    def forward_tree(self, outputs, targets):
        if self.epochs < self.start_epochs:
            return self.criterion(outputs, targets)  # regular xent
        return super().forward_tree(outputs, targets)

    def forward(self, outputs, targets):
        if self.epochs < self.start_epochs:
            return TreeSupLoss.forward(self, outputs, targets)
        return super().forward(outputs, targets)

    def loss_and_grad(self, outputs, targets, grad_scale=1.0):
        if self.epochs < self.start_epochs:      # w_x*CE + w_t*CE through the same fused kernel
            self.assert_output_not_nbdt(outputs)
            xent_weight, tree_weight = self.current_weights()
            handle = self.tree.device_handle(outputs.device.index)
            return _C.soft_tree_loss(handle, outputs, targets, float(xent_weight) + float(tree_weight), 0.0,
                                     grad_scale)
        return super().loss_and_grad(outputs, targets, grad_scale)

    def set_epoch(self, *args, **kwargs):
        super().set_epoch(*args, **kwargs)
        offset = self.epochs - self.start_epochs
        if offset >= 0 and offset % self.update_every_epochs == 0 and self.epochs < self.update_end_epochs:
            import os
            import torch.distributed as dist
            checkpoint_dir = self.checkpoint_path.replace(".pth", "")
            path_graph = os.path.join(checkpoint_dir, f"graph-epoch{self.epochs}.json")
            # One process per GPU (main.py): every rank holds identical weights, but only ONE may write the JSON --
            # concurrent open(path, "w") from N ranks lets a rank read a half-written file.  Rank 0 induces and
            # writes (atomically, Tree.update_from_model), the others wait and then load what it wrote.
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            if not multi or dist.get_rank() == 0:
                self.tree.update_from_model(self.net, self.arch, self.tree.dataset, path_graph=path_graph)
            if multi:
                dist.barrier()
                if dist.get_rank() != 0:
                    self.tree.load_hierarchy(self.tree.dataset, path_graph, self.tree.path_wnids, self.tree.classes)
