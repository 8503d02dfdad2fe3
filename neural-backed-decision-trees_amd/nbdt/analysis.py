"""Analyzers that evaluate the embedded decision rules next to the backbone.

Reference nbdt/analysis.py: the hook protocol of ``Noop`` (:86-127: start/end of epoch, train, test;
``update_batch``), ``DecisionRules`` and its ``HardEmbeddedDecisionRules`` / ``SoftEmbeddedDecisionRules``
subclasses (:204-252), which report the NBDT's accuracy from the backbone's logits.  The rules forward is
the fused HIP kernel (nbdt/model.py here).  The visualisation / entropy-ranking / superclass analyzers of the
reference are presentation tooling outside the hot path and are not built.
"""
import functools

from nbdt import metrics
from nbdt.model import HardEmbeddedDecisionRules as HardRules
from nbdt.model import SoftEmbeddedDecisionRules as SoftRules

__all__ = names = ("Noop", "HardEmbeddedDecisionRules", "SoftEmbeddedDecisionRules")


def start_end_decorator(obj, name):
    start, end = getattr(obj, f"start_{name}"), getattr(obj, f"end_{name}")

    def decorator(f):
        @functools.wraps(f)
        def wrapper(epoch, *args, **kwargs):
            start(epoch)
            out = f(epoch, *args, **kwargs)
            end(epoch)
            return out
        return wrapper
    return decorator


class StartEndContext:
    def __init__(self, obj, name, epoch=0):
        self.obj, self.name, self.epoch = obj, name, epoch

    def __call__(self, epoch):
        self.epoch = epoch
        return self

    def __enter__(self):
        return getattr(self.obj, f"start_{self.name}")(self.epoch)

    def __exit__(self, type, value, traceback):
        getattr(self.obj, f"end_{self.name}")(self.epoch)


class Noop:
    accepts_classes = lambda testset, **kwargs: testset.classes
    name = ""

    def __init__(self, classes=()):
        self.classes = classes
        self.num_classes = len(classes)
        self.epoch = None

    @property
    def epoch_function(self):
        return start_end_decorator(self, "epoch")

    @property
    def train_function(self):
        return start_end_decorator(self, "train")

    @property
    def test_function(self):
        return start_end_decorator(self, "test")

    @property
    def epoch_context(self):
        return StartEndContext(self, "epoch")

    def start_epoch(self, epoch):
        self.epoch = epoch

    def start_train(self, epoch):
        assert epoch == self.epoch

    def update_batch(self, outputs, targets, images=None):
        return self._update_batch(outputs, targets)

    def _update_batch(self, outputs, targets):
        pass

    def end_train(self, epoch):
        assert epoch == self.epoch

    def start_test(self, epoch):
        assert epoch == self.epoch

    def end_test(self, epoch):
        assert epoch == self.epoch

    def end_epoch(self, epoch):
        assert epoch == self.epoch


class DecisionRules(Noop):
    """Generic support for evaluating embedded decision rules (reference :204-237)."""

    accepts_tree = lambda tree, **kwargs: tree
    accepts_dataset = lambda trainset, **kwargs: trainset.__class__.__name__
    accepts_path_graph = True
    accepts_path_wnids = True
    accepts_metric = True
    name = "NBDT"

    def __init__(self, *args, Rules=HardRules, tree=None, metric="top1", **kwargs):
        self.rules = Rules(*args, **kwargs, tree=tree)
        super().__init__(self.rules.tree.classes)
        self.metric = getattr(metrics, metric)()
        self.best_accuracy = 0

    def start_test(self, epoch):
        self.metric.clear()

    def _update_batch(self, outputs, targets):
        outputs = self.rules.forward(outputs)
        self.metric.forward(outputs, targets)
        return round(self.metric.correct / float(self.metric.total), 4) * 100

    def end_test(self, epoch):
        accuracy = round(self.metric.correct / max(self.metric.total, 1) * 100.0, 2)
        self.best_accuracy = max(accuracy, self.best_accuracy)
        print(f"[{self.name}] Accuracy: {accuracy}%, {self.metric.correct}/{self.metric.total} | "
              f"{self.name} Best Accuracy: {self.best_accuracy}%")


class HardEmbeddedDecisionRules(DecisionRules):
    """Evaluation is hard."""
    name = "NBDT-Hard"


class SoftEmbeddedDecisionRules(DecisionRules):
    """Evaluation is soft."""
    name = "NBDT-Soft"

    def __init__(self, *args, Rules=None, **kwargs):
        super().__init__(*args, Rules=SoftRules, **kwargs)
