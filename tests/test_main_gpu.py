"""The main.py-compatible driver end to end on the MI355X: train on synthetic data with a tree loss,
best-accuracy checkpoint in the reference's {"net","acc","epoch"} format, resume + eval with the NBDT
analysis, and loading a checkpoint written from a DataParallel model (`module.` prefix)."""
import importlib.util
import os

import pytest
import torch

import nbdt_path

pytestmark = pytest.mark.gpu

spec = importlib.util.spec_from_file_location("nbdt_main", os.path.join(nbdt_path.PKG_DIR, "main.py"))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)


def test_train_checkpoint_resume_eval(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    common = ("--arch ResNet18 --dataset CIFAR10 --batch-size 64 --synthetic 512 --lr 0.05 "
              "--loss SoftTreeSupLoss --analysis HardEmbeddedDecisionRules").split()
    acc, nbdt_acc = M.main(common + ["--epochs", "7"])
    ck = "checkpoint/ckpt-CIFAR10-ResNet18-lr0.05-induced-ResNet18-SoftTreeSupLoss.pth"
    assert os.path.exists(ck)
    state = torch.load(ck, map_location="cpu")
    assert set(state) == {"net", "acc", "epoch"} and "linear.weight" in state["net"]
    assert acc > 25.0 and nbdt_acc > 20.0          # 10 classes: well above chance after 7 short epochs
    acc2, nbdt2 = M.main(common + ["--resume", "--eval"])
    assert abs(acc2 - state["acc"]) < 1e-6          # the evaluated model IS the checkpointed one
    # a DataParallel-style checkpoint (module. prefix, bare state dict) loads through --path-resume
    torch.save({"module." + k: v for k, v in state["net"].items()}, "dp.pth")
    acc3, _ = M.main(common + ["--resume", "--eval", "--path-resume", "dp.pth"])
    assert abs(acc3 - acc2) < 1e-6


def test_hard_loss_and_plain_cross_entropy_run(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    base = "--arch ResNet18 --dataset CIFAR10 --batch-size 64 --synthetic 256 --lr 0.05 --epochs 5".split()
    acc, _ = M.main(base + ["--loss", "HardTreeSupLoss", "--hierarchy", "induced-ResNet18"])
    assert acc > 15.0
    acc, _ = M.main(base)
    assert acc > 15.0 and os.path.exists("checkpoint/ckpt-CIFAR10-ResNet18-lr0.05.pth")
