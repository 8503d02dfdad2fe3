"""Host logic of the main.py-compatible driver (no GPU): flag surface, checkpoint naming
(reference nbdt/utils.py:266-330) and the MultiStepLR schedule (reference main.py:208-210)."""
import importlib.util
import os

import torch

import nbdt_path

spec = importlib.util.spec_from_file_location("nbdt_main", os.path.join(nbdt_path.PKG_DIR, "main.py"))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)


def test_checkpoint_names_follow_the_reference_convention():
    f = M.generate_checkpoint_fname
    assert f("CIFAR10", "ResNet18") == "ckpt-CIFAR10-ResNet18"
    assert f("CIFAR10", "wrn28_10_cifar10", path_graph="/x/graph-induced-wrn28_10_cifar10.json",
             loss=["SoftTreeSupLoss"]) == "ckpt-CIFAR10-wrn28_10_cifar10-induced-wrn28_10_cifar10-SoftTreeSupLoss"
    # README.md / scripts: TinyImagenet200 wrn28_10 with tsw 10, fine-tune lr 0.01
    assert f("TinyImagenet200", "wrn28_10", path_graph="graph-induced-wrn28_10.json", loss=["SoftTreeSupLoss"],
             tree_supervision_weight=10.0, lr=0.01) == \
        "ckpt-TinyImagenet200-wrn28_10-lr0.01-induced-wrn28_10-SoftTreeSupLoss-tsw10.0"
    assert f("CIFAR100", "ResNet18", name="exp", loss=["HardTreeSupLoss"], path_graph="graph-wordnet.json",
             xent_weight=0.5, tree_supervision_weight_end=5.0) == \
        "ckpt-CIFAR100-ResNet18-exp-wordnet-HardTreeSupLoss-tswe5.0-xw0.5"


def test_lr_schedule_equals_torch_multisteplr():
    for epochs in (7, 10, 200):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=0.1)
        sch = torch.optim.lr_scheduler.MultiStepLR(
            opt, milestones=[int(3 / 7.0 * epochs), int(5 / 7.0 * epochs)])
        for e in range(epochs):
            assert abs(M.multistep_lr(0.1, e, epochs) - sch.get_last_lr()[0]) < 1e-12, (epochs, e)
            opt.step()
            sch.step()


def test_flag_surface_matches_the_reference_recipes():
    p = M.build_parser()
    a = p.parse_args("--dataset CIFAR100 --arch wrn28_10_cifar100 --batch-size 256 --epochs 20 --lr 0.01 "
                     "--loss SoftTreeSupLoss --tsw 10 --hierarchy induced-wrn28_10_cifar100 "
                     "--analysis SoftEmbeddedDecisionRules --resume --eval".split())
    assert a.tree_supervision_weight == 10 and a.loss == ["SoftTreeSupLoss"] and a.resume and a.eval
    assert a.hierarchy == "induced-wrn28_10_cifar100" and a.arch == "wrn28_10_cifar100"


def test_loss_module_owns_its_flags_and_path_defaults():
    """reference nbdt/loss.py:27-91: `add_arguments` defines the weight-schedule / re-induction flags (short aliases
    included) and `set_default_values` resolves --hierarchy / --path-graph / --path-wnids; the driver's parser takes
    its loss flags from there."""
    import argparse
    from nbdt import loss as losses
    p = argparse.ArgumentParser()
    losses.add_arguments(p)
    a = p.parse_args(["--tsw", "10", "--tswe", "1", "--tswp", "2", "--xw", "0.5", "--xwe", "0", "--xwp", "3",
                      "--tse", "5", "--tuene", "9", "--tueve", "2"])
    assert (a.tree_supervision_weight, a.tree_supervision_weight_end, a.tree_supervision_weight_power) == (10, 1, 2)
    assert (a.xent_weight, a.xent_weight_end, a.xent_weight_power) == (0.5, 0, 3)
    assert (a.tree_start_epochs, a.tree_update_end_epochs, a.tree_update_every_epochs) == (5, 9, 2)
    assert p.parse_args([]).tree_supervision_weight == 1
    ns = argparse.Namespace(dataset="CIFAR10", hierarchy="induced-ResNet18", path_graph=None, path_wnids=None)
    losses.set_default_values(ns)
    assert ns.path_graph.endswith("graph-induced-ResNet18.json") and ns.path_wnids.endswith("wnids/CIFAR10.txt")
    ns = argparse.Namespace(dataset="CIFAR10", hierarchy=None, path_graph=None, path_wnids=None)
    losses.set_default_values(ns)
    assert ns.path_graph.endswith("graph-induced.json")
    import pytest
    with pytest.raises(AssertionError, match="Only one"):
        losses.set_default_values(argparse.Namespace(dataset="CIFAR10", hierarchy="induced", path_graph="x.json", path_wnids=None))
    import main as driver
    assert driver.build_parser().parse_args(["--tsw", "3"]).tree_supervision_weight == 3
