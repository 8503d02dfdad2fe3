"""nbdt.analysis: the analyzer hook protocol of the reference's main.py (reference nbdt/analysis.py:81-130, main.py:212-288)
-- call order, epoch checks, the decorator / context-manager wrappers -- on the CPU (no kernel runs in a Noop)."""
import pytest

from nbdt import analysis


class Recorder(analysis.Noop):
    def __init__(self):
        super().__init__(classes=("a", "b", "c"))
        self.calls = []

    def start_epoch(self, epoch):
        super().start_epoch(epoch); self.calls.append(("start_epoch", epoch))

    def end_epoch(self, epoch):
        super().end_epoch(epoch); self.calls.append(("end_epoch", epoch))

    def start_train(self, epoch):
        super().start_train(epoch); self.calls.append(("start_train", epoch, self.phase))

    def end_train(self, epoch):
        super().end_train(epoch); self.calls.append(("end_train", epoch, self.phase))

    def start_test(self, epoch):
        super().start_test(epoch); self.calls.append(("start_test", epoch, self.phase))

    def end_test(self, epoch):
        super().end_test(epoch); self.calls.append(("end_test", epoch, self.phase))

    def _update_batch(self, outputs, targets):
        self.calls.append(("batch", outputs, targets))
        return len(self.calls)


def test_names_and_classes():
    assert analysis.names == ("Noop", "HardEmbeddedDecisionRules", "SoftEmbeddedDecisionRules")
    for n in analysis.names:
        assert issubclass(getattr(analysis, n), analysis.Noop)
    a = analysis.Noop(classes=["x", "y"])
    assert a.num_classes == 2 and a.classes == ("x", "y") and a.update_batch(None, None, None) is None
    assert analysis.HardEmbeddedDecisionRules.name == "NBDT-Hard" and analysis.SoftEmbeddedDecisionRules.name == "NBDT-Soft"
    assert analysis.DecisionRules.accepts_path_graph and analysis.DecisionRules.accepts_metric


def test_wrappers_drive_the_hooks_in_the_reference_order():
    r = Recorder()

    @r.train_function
    def train(epoch, n):
        for i in range(n):
            r.update_batch(f"z{i}", f"y{i}", None)
        return "trained"

    @r.test_function
    def test(epoch):
        return r.update_batch("zt", "yt", None)

    for epoch in (0, 1):
        with r.epoch_context(epoch):
            assert train(epoch, 2) == "trained"
            assert test(epoch) is not None
    e0 = [c[0] for c in r.calls[:len(r.calls) // 2]]
    assert e0 == ["start_epoch", "start_train", "batch", "batch", "end_train", "start_test", "batch", "end_test",
                  "end_epoch"]
    assert r.calls[1] == ("start_train", 0, "train") and r.calls[4] == ("end_train", 0, None)
    assert r.calls[5] == ("start_test", 0, "test") and r.epoch == 1

    @r.epoch_function
    def whole(epoch):
        return epoch * 2
    assert whole(5) == 10 and r.calls[-2:] == [("start_epoch", 5), ("end_epoch", 5)]


def test_a_hook_for_another_epoch_is_an_error():
    r = analysis.Noop()
    r.start_epoch(3)
    r.start_test(3)
    with pytest.raises(AssertionError):
        r.end_test(4)
    with pytest.raises(ValueError):
        analysis.DecisionRules.__init__(analysis.DecisionRules.__new__(analysis.DecisionRules), dataset="CIFAR10",
                                        hierarchy="induced-ResNet18", metric="top3")


def test_an_exception_inside_a_phase_still_ends_it():
    """the reference's StartEndContext.__exit__ (analysis.py:77-78) calls end_* whatever happened in the body."""
    r = Recorder()
    with pytest.raises(RuntimeError, match="boom"):
        with r.epoch_context(2):
            @r.train_function
            def train(epoch):
                raise RuntimeError("boom")
            train(2)
    assert [c[0] for c in r.calls] == ["start_epoch", "start_train", "end_train", "end_epoch"]
    assert r.phase is None and r.calls[2] == ("end_train", 2, None)


def test_decision_rules_start_test_adopts_the_epoch_like_the_reference():
    """Reference analysis.py:221-222: DecisionRules.start_test does not assert the epoch (an eval-only driver never called
    start_epoch).  No kernel runs here: the object is built without its rules."""
    d = analysis.DecisionRules.__new__(analysis.DecisionRules)
    analysis.Noop.__init__(d, classes=("a", "b"))
    d.best_accuracy, d.verbose = 0.0, False
    d.start_test(7)                      # no enclosing start_epoch: must not raise
    assert d.epoch == 7 and d.phase == "test"
    d.end_test(7)
