"""Pins the CPU oracle (oracle/nbdt_oracle.py) to the UNMODIFIED reference.

The fixtures tests/golden/rules_*.npz were produced by tests/golden/make_golden.py, which imports
/root/reference and records its outputs; they are the only golden vectors for this path (the
reference's own tests are smoke tests).  Tolerances are the ones SURVEY 8c
states for the fp32 rules layer: rtol 2e-5 / atol 1e-6 on P, 1e-5 rel on the loss, 1e-6 abs on
dL/dz; every integer output (argmax, hard prediction, per-node argmax, tree maps) is bit-exact.
"""
import os

import numpy as np
import pytest

import nbdt_oracle as O
from conftest import GOLDEN_CASES


def _load(tag, golden_dir, pkg_dir):
    ds, h = GOLDEN_CASES[tag]
    g = np.load(os.path.join(golden_dir, f"rules_{tag}.npz"))
    t = O.OracleTree(*O.default_paths(ds, h, pkg_dir))
    return g, t


@pytest.mark.parametrize("tag", list(GOLDEN_CASES))
def test_tree_maps_match_reference(tag, golden_dir, pkg_dir):
    g, t = _load(tag, golden_dir, pkg_dir)
    assert list(g["tree_inode_wnids"]) == t.inode_wnids
    assert str(g["tree_root"]) == t.inode_wnids[t.root]
    assert list(g["tree_wnids_leaves"]) == t.wnids_leaves
    child_wnid = [c for n in t.children for c in n]
    assert list(g["tree_child_wnid"]) == child_wnid
    off = np.cumsum([0] + [len(n) for n in t.children])
    assert np.array_equal(off, g["tree_child_off"])
    flat = [c for n in t.child_classes for k in n for c in k]
    assert np.array_equal(np.array(flat, dtype=np.int32), g["tree_slot_cls"])


@pytest.mark.parametrize("tag", list(GOLDEN_CASES))
def test_rules_match_reference(tag, golden_dir, pkg_dir):
    g, t = _load(tag, golden_dir, pkg_dir)
    z = g["z"]
    outs = O.node_outputs(t, z)
    P = O.soft_forward(t, z, outs)
    np.testing.assert_allclose(P, g["soft_P"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(P.sum(1), 1.0, atol=1e-5)
    assert np.array_equal(P.argmax(1), g["soft_P"].argmax(1))
    pred, dec = O.hard_forward(t, z, outs, with_decisions=True)
    assert np.array_equal(pred, g["hard_pred"])
    assert np.array_equal(O.hard_onehot(t, pred).argmax(1), g["hard_pred"])
    nl = np.concatenate([o["logits"] for o in outs], 1)
    npb = np.concatenate([o["probs"] for o in outs], 1)
    np.testing.assert_allclose(nl, g["node_logits"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(npb, g["node_probs"], rtol=2e-5, atol=1e-6)
    assert np.array_equal(np.stack([o["preds"] for o in outs], 1), g["node_preds"])
    np.testing.assert_allclose(np.stack([o["entropy"] for o in outs], 1), g["node_entropy"],
                               rtol=1e-4, atol=1e-6)
    # decisions of the first 4 samples (node path, child index, prob, entropy)
    for i in range(4):
        steps = dec[i]
        n = int((g["dec_path"][i] >= 0).sum())
        assert len(steps) == n
        for j, (node, k, prob, ent) in enumerate(steps):
            assert node == g["dec_path"][i, j] and k == g["dec_next"][i, j]
            assert abs(prob - g["dec_prob"][i, j]) < 1e-6
            assert abs(ent - g["dec_entropy"][i, j]) < 1e-5


@pytest.mark.parametrize("tag", list(GOLDEN_CASES))
def test_loss_and_grad_match_reference(tag, golden_dir, pkg_dir):
    g, t = _load(tag, golden_dir, pkg_dir)
    z, y = g["z"], g["y"]
    loss, dz = O.soft_tree_sup_loss(t, z, y)
    assert abs(loss - g["loss"]) <= 1e-5 * abs(g["loss"])
    np.testing.assert_allclose(dz, g["dz"], atol=1e-6, rtol=0)
    loss_w, dz_w = O.soft_tree_sup_loss(t, z, y, w_xent=0.5, w_tree=10.0)
    assert abs(loss_w - g["loss_w"]) <= 1e-5 * abs(g["loss_w"])
    np.testing.assert_allclose(dz_w, g["dz_w"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(O.rules_backward(t, z, g["gP"]), g["dz_rules"], atol=2e-6, rtol=1e-5)


def test_all_zero_logits_tie_goes_to_child_zero(golden_dir, pkg_dir):
    # all-zero logits on the CIFAR10 induced-wrn tree -> hard pred class 4
    _, t = _load("cifar10_wrn", golden_dir, pkg_dir)
    pred = O.hard_forward(t, np.zeros((3, 10), dtype=np.float32))
    assert list(pred) == [4, 4, 4]


def test_weight_schedule():
    # nbdt/loss.py:187-189
    assert O.tree_weight(1.0, 1.0, 5.0) == 5.0
    assert O.tree_weight(0.0, 1.0, 5.0) == 1.0
    assert abs(O.tree_weight(0.5, 1.0, 5.0, power=2) - (0.75 * 1 + 0.25 * 5)) < 1e-12


@pytest.mark.parametrize("tag", list(GOLDEN_CASES))
def test_hard_loss_and_grad_match_reference(tag, golden_dir, pkg_dir):
    """HardTreeSupLoss (nbdt/loss.py:212-257) incl. its tree_supervision_weight-applied-twice quirk."""
    g, t = _load(tag, golden_dir, pkg_dir)
    z, y = g["z"], g["y"]
    loss, dz = O.hard_tree_sup_loss(t, z, y)
    assert abs(loss - g["hloss"]) <= 1e-5 * abs(g["hloss"])
    np.testing.assert_allclose(dz, g["hdz"], atol=1e-6, rtol=0)
    loss_w, dz_w = O.hard_tree_sup_loss(t, z, y, w_xent=0.5, tree_supervision_weight=10.0)
    assert abs(loss_w - g["hloss_w"]) <= 1e-5 * abs(g["hloss_w"])
    np.testing.assert_allclose(dz_w, g["hdz_w"], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("tag,dataset,num_classes", [("cifar10", "CIFAR10", 10), ("tiny200", "TinyImagenet200", 200)])
def test_backbone_oracle_resnet18_is_the_reference_resnet18(tag, dataset, num_classes, golden_dir, pkg_dir):
    """Pins oracle/torch_models.ResNet18 to the reference's own class (nbdt/models/resnet.py:171-179, recorded by
    make_golden.run_backbone): same seed -> the same parameters in the same state-dict order, the same train-mode
    logits, the same SoftTreeSupLoss (numpy oracle on the oracle logits) and the same per-parameter gradient norms
    and BatchNorm running statistics."""
    import torch
    import torch_models as TM
    g = np.load(os.path.join(golden_dir, f"backbone_resnet18_{tag}.npz"))
    torch.manual_seed(int(g["seed"]))
    net = TM.ResNet18(num_classes=num_classes)
    net.train()
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["param_sums"], rtol=0, atol=1e-9)
    x, y = torch.from_numpy(g["x"]), g["y"]
    z = net(x)
    np.testing.assert_allclose(z.detach().numpy(), g["logits"], rtol=1e-5, atol=1e-5)
    otree = O.OracleTree(*O.default_paths(dataset, "induced-ResNet18", pkg_dir))
    loss, dz = O.soft_tree_sup_loss(otree, z.detach().numpy(), y)
    assert abs(loss - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    z.backward(torch.from_numpy(dz))
    names = [n for n, _ in net.named_parameters()]
    assert names == list(g["grad_names"])
    gn = np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()])
    np.testing.assert_allclose(gn, g["grad_norms"], rtol=2e-4, atol=1e-7)
    sd = net.state_dict()
    np.testing.assert_allclose(sd["bn1.running_mean"].numpy(), g["bn1_running_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sd["layer4.1.bn2.running_var"].numpy(), g["last_running_var"], rtol=1e-5, atol=1e-6)


def test_wrn_oracle_keys_are_the_released_checkpoint_keys():
    """WRN-28-10 stays parity-unpinned (pytorchcv absent).  What the reference does pin is the key the released
    checkpoints are read with: `output.weight` is the classifier key nbdt/graph.py:391 looks up in a
    wrn28_10_cifar10 checkpoint (nbdt/model.py:32-35 lists those checkpoints), and the canonical WRN-28-10 has
    36,454,832 conv weights in 28 convolutions."""
    import torch_models as TM
    sd = TM.WRN(10, 28, 10).state_dict()
    assert sd["output.weight"].shape == (10, 640) and sd["output.bias"].shape == (10,)
    conv = [k for k, v in sd.items() if v.dim() == 4]
    assert len(conv) == 28 and sum(sd[k].numel() for k in conv) == 36454832
    assert "features.init_block.weight" in sd and "features.stage3.unit4.body.conv2.conv.weight" in sd
    assert "features.stage2.unit1.identity_conv.weight" in sd and "features.post_activ.bn.running_var" in sd
