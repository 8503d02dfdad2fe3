"""Per-epoch analyzers: the second caller of the decision-rules layer.

API surface of the reference's ``nbdt/analysis.py`` for the hot path -- the hook protocol its ``main.py`` drives
(reference nbdt/analysis.py:81-130: ``start_epoch / start_train / update_batch / end_train / start_test / end_test /
end_epoch``, the ``epoch_context`` / ``*_function`` wrappers, main.py:212-288) and the two analyzers that run the
embedded decision rules on the backbone's logits (``HardEmbeddedDecisionRules`` / ``SoftEmbeddedDecisionRules``,
reference :204-252), so that ``--analysis <name>`` and a user's own subclass keep working.  The presentation analyzers
(confusion matrices, entropy rankings, image dumps) are out of scope.

What is behind the surface is this repository's: the rules are the fused HIP kernels of ``nbdt.model`` (one launch per
batch), and the hit counters live ON THE DEVICE -- ``update_batch`` enqueues work and returns without a host
synchronisation; the only transfer is the one ``end_test`` / ``accuracy()`` needs to print.  (The reference computes a
running accuracy on the host after every batch; pass ``sync_every_batch=True`` to get that return value.)
"""
import contextlib
import functools

import torch

from nbdt.model import HardEmbeddedDecisionRules as _HardRules
from nbdt.model import SoftEmbeddedDecisionRules as _SoftRules

__all__ = names = ("Noop", "HardEmbeddedDecisionRules", "SoftEmbeddedDecisionRules")
_TOPK = {"top1": 1, "top2": 2, "top5": 5, "top10": 10}      # the reference's --metric names


def add_arguments(parser):
    """The reference registers flags of its presentation analyzers here; the analyzers of this module have none."""


class _Phase:
    """``start_<name>(epoch)`` ... ``end_<name>(epoch)`` of one analyzer as a decorator factory and a context manager."""

    def __init__(self, owner, name):
        self._begin = getattr(owner, "start_" + name)
        self._finish = getattr(owner, "end_" + name)

    def wrap(self, fn):
        @functools.wraps(fn)
        def bracketed(epoch, *args, **kwargs):
            with self(epoch):
                return fn(epoch, *args, **kwargs)
        return bracketed

    @contextlib.contextmanager
    def __call__(self, epoch):
        self._begin(epoch)
        try:
            yield
        finally:                 # like the reference's StartEndContext.__exit__ (analysis.py:77-78): end_* runs even when
            self._finish(epoch)  # the body raises, so the analyzer's phase / epoch state never goes stale


class Noop:
    """Does nothing at every hook; the base class (and the default ``--analysis``).

    A training driver calls, per epoch:  start_epoch, [start_train, update_batch x N, end_train],
    [start_test, update_batch x M, end_test], end_epoch -- each with the epoch it belongs to."""

    accepts_classes = lambda testset, **kwargs: testset.classes     # noqa: E731  (generate_kwargs protocol)
    name = "Noop"

    def __init__(self, classes=()):
        self.classes = tuple(classes)
        self.num_classes = len(self.classes)
        self.epoch = None
        self.phase = None          # None | "train" | "test"
        self.verbose = True        # end-of-pass reports are printed (a multi-rank driver leaves it on for one rank)

    # ---- wrappers a driver can use instead of calling the hooks itself
    @property
    def epoch_function(self):
        return _Phase(self, "epoch").wrap

    @property
    def train_function(self):
        return _Phase(self, "train").wrap

    @property
    def test_function(self):
        return _Phase(self, "test").wrap

    @property
    def epoch_context(self):
        return _Phase(self, "epoch")

    # ---- hooks
    def start_epoch(self, epoch):
        self.epoch = epoch

    def end_epoch(self, epoch):
        self._same_epoch(epoch)

    def start_train(self, epoch):
        self._same_epoch(epoch)
        self.phase = "train"

    def end_train(self, epoch):
        self._same_epoch(epoch)
        self.phase = None

    def start_test(self, epoch):
        self._same_epoch(epoch)
        self.phase = "test"

    def end_test(self, epoch):
        self._same_epoch(epoch)
        self.phase = None

    def update_batch(self, outputs, targets, images=None):
        """outputs: the backbone's logits [B, classes] (device), targets [B]; returns a per-batch statistic or None."""
        return self._update_batch(outputs, targets)

    def _update_batch(self, outputs, targets):
        return None

    def _same_epoch(self, epoch):
        if epoch != self.epoch:
            raise AssertionError(f"hook called for epoch {epoch} inside epoch {self.epoch}")


class DecisionRules(Noop):
    """Accuracy of embedded decision rules applied to the backbone's logits, over a test pass."""

    accepts_tree = lambda tree, **kwargs: tree                                        # noqa: E731
    accepts_dataset = lambda trainset, **kwargs: trainset.__class__.__name__          # noqa: E731
    accepts_path_graph = True
    accepts_path_wnids = True
    accepts_metric = True
    name = "NBDT"

    def __init__(self, *args, Rules=_HardRules, tree=None, metric="top1", sync_every_batch=False, **kwargs):
        self.rules = Rules(*args, tree=tree, **kwargs)
        super().__init__(self.rules.tree.classes)
        if metric not in _TOPK:
            raise ValueError(f"metric must be one of {sorted(_TOPK)}, got {metric!r}")
        self.k = _TOPK[metric]
        self.sync_every_batch = bool(sync_every_batch)
        self.best_accuracy = 0.0
        self._hits = None          # device scalar, created on the first batch's device
        self._seen = 0

    # device-side counting: nothing here waits for the GPU
    def _count(self, scores, targets):
        k = min(self.k, scores.shape[1])
        if k == 1:
            hit = scores.argmax(dim=1) == targets
        else:
            hit = (scores.topk(k, dim=1).indices == targets[:, None]).any(dim=1)
        if self._hits is None or self._hits.device != hit.device:
            self._hits = torch.zeros((), dtype=torch.long, device=hit.device)
        self._hits += hit.sum()
        self._seen += int(targets.shape[0])

    @property
    def correct(self):
        return 0 if self._hits is None else int(self._hits)      # (host transfer)

    @property
    def total(self):
        return self._seen

    def accuracy(self):
        """Percent of the samples counted since the last start_test whose target is among the rules' top-k."""
        return 100.0 * self.correct / max(self._seen, 1)

    def start_test(self, epoch):
        # The reference's override (analysis.py:221-222) does not check the epoch: a driver that evaluates without an
        # enclosing start_epoch (an eval-only run) must not raise here.  Adopt the epoch instead of asserting it.
        self.epoch = epoch
        super().start_test(epoch)
        self._hits, self._seen = None, 0

    def _update_batch(self, outputs, targets):
        with torch.no_grad():
            self._count(self.rules.forward(outputs), targets)
        return round(self.accuracy(), 2) if self.sync_every_batch else None

    def end_test(self, epoch):
        super().end_test(epoch)
        acc = round(self.accuracy(), 2)
        self.best_accuracy = max(self.best_accuracy, acc)
        if self.verbose:
            print(f"[{self.name}] rules accuracy {acc}% ({self.correct} of {self.total}); best so far {self.best_accuracy}%")


class HardEmbeddedDecisionRules(DecisionRules):
    """Greedy root-to-leaf walk (argmax at every node): one-hot scores of the predicted leaf."""

    name = "NBDT-Hard"


class SoftEmbeddedDecisionRules(DecisionRules):
    """Path-probability product over the whole tree."""

    name = "NBDT-Soft"

    def __init__(self, *args, Rules=None, **kwargs):
        super().__init__(*args, Rules=_SoftRules, **kwargs)
