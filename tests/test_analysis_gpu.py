"""nbdt.analysis on the MI355X: the rules analyzers (the second caller of the decision-rules kernels, SURVEY.md
section 2 row 14) count, on the device, exactly what the numpy oracle's rules give on the engine's own logits."""
import os

import numpy as np
import pytest
import torch

import nbdt_oracle as O
import nbdt_path

pytestmark = pytest.mark.gpu

from nbdt import analysis  # noqa: E402
from nbdt import engine as E  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("dataset,hierarchy,classes,size,make", [
    ("CIFAR10", "induced-ResNet18", 10, 32, lambda: E.ResNetEngine(10, device=DEV, seed=3)),
    ("TinyImagenet200", "induced-ResNet18", 200, 64, lambda: E.ResNetEngine(200, device=DEV, seed=3)),
    ("CIFAR100", "induced-wrn28_10_cifar100", 100, 32,
     lambda: E.WRNEngine(num_classes=100, blocks=10, width_factor=2, device=DEV, seed=3)),
])
def test_rules_analyzers_equal_the_oracle_on_the_engines_logits(dataset, hierarchy, classes, size, make):
    eng = make()
    otree = O.OracleTree(*O.default_paths(dataset, hierarchy, os.path.join(nbdt_path.PKG_DIR, "nbdt")))
    hard = analysis.HardEmbeddedDecisionRules(dataset=dataset, hierarchy=hierarchy)
    soft = analysis.SoftEmbeddedDecisionRules(dataset=dataset, hierarchy=hierarchy)
    soft5 = analysis.SoftEmbeddedDecisionRules(dataset=dataset, hierarchy=hierarchy, metric="top5", sync_every_batch=True)
    assert hard.classes == soft.classes and hard.num_classes == classes
    g = torch.Generator().manual_seed(11)
    want = {"hard": 0, "soft": 0, "soft5": 0}
    seen = 0
    for a in (hard, soft, soft5):
        a.start_epoch(2)
        a.start_test(2)
    for batch in (48, 48, 17):                                  # ragged last batch
        x = torch.randn(batch, 3, size, size, generator=g).to(DEV)
        y = torch.randint(0, classes, (batch,), generator=g)
        z = eng.forward(x, training=False)
        zn = z.float().cpu().numpy()
        want["hard"] += int((O.hard_forward(otree, zn) == y.numpy()).sum())
        P = O.soft_forward(otree, zn)
        want["soft"] += int((P.argmax(1) == y.numpy()).sum())
        want["soft5"] += int((np.argsort(-P, axis=1, kind="stable")[:, :5] == y.numpy()[:, None]).any(1).sum())
        seen += batch
        assert hard.update_batch(z, y.to(DEV), x) is None        # no host synchronisation by default
        assert soft.update_batch(z, y.to(DEV), x) is None
        assert isinstance(soft5.update_batch(z, y.to(DEV), x), float)
    assert (hard.total, soft.total) == (seen, seen)
    assert hard.correct == want["hard"]                          # greedy walk: bit-exact decisions
    assert abs(soft.correct - want["soft"]) <= 1                 # (path probabilities agree to 2e-5: a near-tie may flip)
    assert abs(soft5.correct - want["soft5"]) <= 1
    for a in (hard, soft):
        assert abs(a.accuracy() - 100.0 * a.correct / seen) < 1e-9
        a.end_test(2)
        a.end_epoch(2)
        assert a.best_accuracy == round(100.0 * a.correct / seen, 2)
    hard.start_epoch(3)
    hard.start_test(3)                                           # a new test pass starts from zero
    assert hard.total == 0 and hard.correct == 0
