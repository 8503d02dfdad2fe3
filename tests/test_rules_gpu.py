"""Parity of the HIP rules layer / SoftTreeSupLoss (through the C-ABI) against the CPU oracle and
against the reference's golden vectors.  Integer outputs are bit-exact; fp32 tolerances are the
ones stated in SURVEY.md 8c: P rtol 2e-5 / atol 1e-6, loss 1e-5 rel, dL/dz 1e-6 abs."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import nbdt_oracle as O
from conftest import GOLDEN_CASES

pytestmark = pytest.mark.gpu

from nbdt import _C  # noqa: E402
from nbdt.loss import HardTreeSupLoss, SoftTreeLoss, SoftTreeSupLoss  # noqa: E402
from nbdt.model import (HardEmbeddedDecisionRules, HardNBDT, SoftEmbeddedDecisionRules,  # noqa: E402
                        SoftNBDT)
from nbdt.tree import Tree  # noqa: E402

DEV = "cuda:0"


def _case(tag, golden_dir, pkg_dir):
    ds, h = GOLDEN_CASES[tag]
    g = np.load(os.path.join(golden_dir, f"rules_{tag}.npz"))
    return g, Tree(ds, hierarchy=h), O.OracleTree(*O.default_paths(ds, h, pkg_dir)), ds, h


@pytest.mark.parametrize("tag", list(GOLDEN_CASES))
def test_golden_inputs(tag, golden_dir, pkg_dir):
    g, tree, otree, ds, h = _case(tag, golden_dir, pkg_dir)
    z = torch.from_numpy(g["z"]).to(DEV)
    y = torch.from_numpy(g["y"]).to(DEV)
    handle = tree.device_handle(0)

    P = _C.soft_forward(handle, z).cpu().numpy()
    outs = O.node_outputs(otree, g["z"])
    Po = O.soft_forward(otree, g["z"], outs)
    np.testing.assert_allclose(P, Po, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(P, g["soft_P"], rtol=2e-5, atol=1e-6)
    assert np.array_equal(P.argmax(1), g["soft_P"].argmax(1))

    pred, onehot, (pn, pc, pp, pe) = _C.hard_forward(handle, z, want_onehot=True, want_decisions=True)
    pred = pred.cpu().numpy()
    assert np.array_equal(pred, O.hard_forward(otree, g["z"], outs))       # bit-exact vs oracle
    assert np.array_equal(pred, g["hard_pred"])                             # and vs the reference
    oh = onehot.cpu().numpy()
    assert np.array_equal(oh, np.eye(oh.shape[1], dtype=np.float32)[pred])
    D = min(pn.shape[1], 32)
    assert np.array_equal(pn.cpu().numpy()[:4, :D], g["dec_path"][:, :D])
    assert np.array_equal(pc.cpu().numpy()[:4, :D], g["dec_next"][:, :D])
    np.testing.assert_allclose(pp.cpu().numpy()[:4, :D], g["dec_prob"][:, :D], atol=1e-6)
    np.testing.assert_allclose(pe.cpu().numpy()[:4, :D], g["dec_entropy"][:, :D], atol=1e-5)
    assert (g["dec_path"][:, D:] < 0).all()

    for (wx, wt, kl, kd) in [(1.0, 1.0, "loss", "dz"), (0.5, 10.0, "loss_w", "dz_w")]:
        loss, gz = _C.soft_tree_loss(handle, z, y, wx, wt)
        lo, dzo = O.soft_tree_sup_loss(otree, g["z"], g["y"], wx, wt)
        assert abs(loss.item() - lo) <= 1e-5 * abs(lo)
        assert abs(loss.item() - g[kl]) <= 1e-5 * abs(g[kl])
        np.testing.assert_allclose(gz.cpu().numpy(), dzo, atol=1e-6, rtol=0)
        np.testing.assert_allclose(gz.cpu().numpy(), g[kd], atol=1e-6, rtol=0)

    gz = _C.soft_backward(handle, z, torch.from_numpy(g["gP"]).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(gz, g["dz_rules"], atol=2e-6, rtol=1e-5)

    # HardTreeSupLoss (nbdt/loss.py:212-257): fused kernel vs oracle vs the reference's autograd
    N = len(tree.inodes)
    for (wx, tsw, kl, kd) in [(1.0, 1.0, "hloss", "hdz"), (0.5, 10.0, "hloss_w", "hdz_w")]:
        loss, gz = _C.hard_tree_loss(handle, z, y, wx, tsw * tsw * 2.0 / N)
        lo, dzo = O.hard_tree_sup_loss(otree, g["z"], g["y"], wx, tsw)
        assert abs(loss.item() - lo) <= 1e-5 * abs(lo)
        assert abs(loss.item() - g[kl]) <= 1e-5 * abs(g[kl])
        np.testing.assert_allclose(gz.cpu().numpy(), dzo, atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(gz.cpu().numpy(), g[kd], atol=2e-6, rtol=1e-5)
        crit = HardTreeSupLoss(dataset=ds, criterion=nn.CrossEntropyLoss(), tree=tree,
                               tree_supervision_weight=tsw, xent_weight=wx)
        zz = z.clone().requires_grad_(True)
        lm = crit(zz, y)
        lm.backward()
        assert abs(lm.item() - g[kl]) <= 1e-5 * abs(g[kl])
        np.testing.assert_allclose(zz.grad.cpu().numpy(), g[kd], atol=2e-6, rtol=1e-5)

    logits, probs, preds, ent = (t.cpu().numpy() for t in _C.node_outputs(handle, z))
    assert np.array_equal(logits, np.concatenate([o["logits"] for o in outs], 1))  # same fp32 order
    np.testing.assert_allclose(probs, g["node_probs"], rtol=2e-5, atol=1e-6)
    assert np.array_equal(preds, g["node_preds"])
    np.testing.assert_allclose(ent, g["node_entropy"], rtol=1e-4, atol=1e-6)


# BASELINE.json shapes: (B, dataset, hierarchy)
FULL = [(512, "CIFAR10", "induced-wrn28_10_cifar10"), (1024, "CIFAR100", "induced-wrn28_10_cifar100"),
        (1024, "TinyImagenet200", "induced-ResNet18"), (256, "Imagenet1000", "induced-efficientnet_b7b")]


@pytest.mark.parametrize("B,ds,h", FULL)
def test_full_size_vs_oracle(B, ds, h, pkg_dir):
    tree = Tree(ds, hierarchy=h)
    otree = O.OracleTree(*O.default_paths(ds, h, pkg_dir))
    C = len(tree.classes)
    gen = torch.Generator().manual_seed(B + C)
    z = torch.randn(B, C, generator=gen) * 3
    y = torch.randint(0, C, (B,), generator=gen)
    zd, yd = z.to(DEV), y.to(DEV)
    handle = tree.device_handle(0)
    outs = O.node_outputs(otree, z.numpy())
    P = _C.soft_forward(handle, zd).cpu().numpy()
    np.testing.assert_allclose(P, O.soft_forward(otree, z.numpy(), outs), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(P.sum(1), 1.0, atol=2e-5)          # size-independent property
    pred = _C.hard_forward(handle, zd, want_onehot=False)[0].cpu().numpy()
    assert np.array_equal(pred, O.hard_forward(otree, z.numpy(), outs))
    loss, gz = _C.soft_tree_loss(handle, zd, yd, 1.0, 1.0)
    lo, dzo = O.soft_tree_sup_loss(otree, z.numpy(), y.numpy())
    assert abs(loss.item() - lo) <= 1e-5 * abs(lo)
    np.testing.assert_allclose(gz.cpu().numpy(), dzo, atol=1e-6, rtol=0)
    # gradient rows of a softmax-CE sum to ~0 (both terms): property check at full size
    assert np.abs(gz.cpu().numpy().sum(1)).max() < 1e-6
    hl, hgz = _C.hard_tree_loss(handle, zd, yd, 1.0, 2.0 / len(tree.inodes))
    hlo, hdzo = O.hard_tree_sup_loss(otree, z.numpy(), y.numpy())
    assert abs(hl.item() - hlo) <= 1e-5 * abs(hlo)
    np.testing.assert_allclose(hgz.cpu().numpy(), hdzo, atol=1e-6, rtol=1e-5)
    assert np.abs(hgz.cpu().numpy().sum(1)).max() < 1e-6


def _caterpillar(tmp_path, C):
    """Every inner node keeps one leaf and hands the rest down: depth C-1, sum of leaf depths ~ C*C/2."""
    import json
    leaves = [f"f{i:08d}" for i in range(C)]
    inner = [f"n{i:08d}" for i in range(C - 1)]
    links = []
    for i in range(C - 1):
        links.append({"source": inner[i], "target": leaves[i]})
        links.append({"source": inner[i], "target": inner[i + 1] if i + 1 < C - 1 else leaves[C - 1]})
    pg, pw = str(tmp_path / f"cat{C}.json"), str(tmp_path / f"cat{C}.txt")
    with open(pg, "w") as f:
        json.dump({"directed": True, "multigraph": False, "graph": {},
                   "nodes": [{"id": w} for w in inner + leaves], "links": links}, f)
    with open(pw, "w") as f:
        f.write("\n".join(leaves) + "\n")
    return pg, pw, leaves


@pytest.mark.parametrize("C", [640, 900])
def test_deep_hierarchy_without_lds_stage(C, tmp_path):
    """A hierarchy whose leaf-depth sum does not fit the LDS staging row (csrc/rules.hip: TreeView::staged = 0)
    takes the direct-indexed chains: same arithmetic contract."""
    pg, pw, leaves = _caterpillar(tmp_path, C)
    tree = Tree(None, path_graph=pg, path_wnids=pw, classes=leaves)
    otree = O.OracleTree(pg, pw)
    assert int(tree.flat.slot_off[-1]) * 4 > 160 * 1024
    gen = torch.Generator().manual_seed(C)
    B = 6
    z = torch.randn(B, C, generator=gen) * 3
    y = torch.randint(0, C, (B,), generator=gen)
    zd, yd = z.to(DEV), y.to(DEV)
    handle = tree.device_handle(0)
    outs = O.node_outputs(otree, z.numpy())
    logits, probs, preds, ent = (t.cpu().numpy() for t in _C.node_outputs(handle, zd))
    assert np.array_equal(logits, np.concatenate([o["logits"] for o in outs], 1))
    assert np.array_equal(preds, np.stack([o["preds"] for o in outs], 1))
    P = _C.soft_forward(handle, zd).cpu().numpy()
    np.testing.assert_allclose(P, O.soft_forward(otree, z.numpy(), outs), rtol=2e-4, atol=1e-6)  # 639 factors
    pred = _C.hard_forward(handle, zd, want_onehot=False)[0].cpu().numpy()
    assert np.array_equal(pred, O.hard_forward(otree, z.numpy(), outs))
    loss, gz = _C.soft_tree_loss(handle, zd, yd, 1.0, 1.0)
    lo, dzo = O.soft_tree_sup_loss(otree, z.numpy(), y.numpy())
    assert abs(loss.item() - lo) <= 1e-5 * abs(lo)
    np.testing.assert_allclose(gz.cpu().numpy(), dzo, atol=1e-6, rtol=0)
    hl, hgz = _C.hard_tree_loss(handle, zd, yd, 1.0, 2.0 / len(tree.inodes))
    hlo, hdzo = O.hard_tree_sup_loss(otree, z.numpy(), y.numpy())
    assert abs(hl.item() - hlo) <= 1e-5 * abs(hlo)
    np.testing.assert_allclose(hgz.cpu().numpy(), hdzo, atol=1e-6, rtol=1e-5)


def test_edge_cases(pkg_dir):
    tree = Tree("CIFAR10", hierarchy="induced-wrn28_10_cifar10")
    otree = O.OracleTree(*O.default_paths("CIFAR10", "induced-wrn28_10_cifar10", pkg_dir))
    handle = tree.device_handle(0)
    # empty batch
    z0 = torch.empty(0, 10, device=DEV)
    assert _C.soft_forward(handle, z0).shape == (0, 10)
    assert _C.hard_forward(handle, z0)[0].shape == (0,)
    # all-zero logits: every node ties -> child 0 -> class 4
    z = torch.zeros(3, 10, device=DEV)
    assert _C.hard_forward(handle, z)[0].tolist() == [4, 4, 4]
    # B = 1 and ragged batch (not a multiple of the samples-per-block)
    for B in (1, 5, 7):
        zz = torch.randn(B, 10, generator=torch.Generator().manual_seed(B))
        P = _C.soft_forward(handle, zz.to(DEV)).cpu().numpy()
        np.testing.assert_allclose(P, O.soft_forward(otree, zz.numpy()), rtol=2e-5, atol=1e-6)
    # row stride > C (a column slice of a wider matrix) is consumed in place
    wide = torch.randn(9, 16, generator=torch.Generator().manual_seed(3)).to(DEV)
    view = wide[:, :10]
    P = _C.soft_forward(handle, view).cpu().numpy()
    np.testing.assert_allclose(P, O.soft_forward(otree, view.cpu().numpy()), rtol=2e-5, atol=1e-6)
    # bf16 / fp16 logits are up-cast on load, P stays fp32 (boundary contract 7)
    for dt in (torch.bfloat16, torch.float16):
        zb = torch.randn(6, 10, generator=torch.Generator().manual_seed(9)).to(dt)
        P = _C.soft_forward(handle, zb.to(DEV))
        assert P.dtype == torch.float32
        np.testing.assert_allclose(P.cpu().numpy(), O.soft_forward(otree, zb.float().numpy()),
                                   rtol=2e-5, atol=1e-6)
    # huge logits do not overflow
    zh = torch.zeros(2, 10); zh[0, 3] = 1e4; zh[1, 7] = -1e4
    P = _C.soft_forward(handle, zh.to(DEV)).cpu().numpy()
    assert np.isfinite(P).all() and P[0].argmax() == 3
    # invalid label -> NaN loss (loud), not silent garbage
    loss, _ = _C.soft_tree_loss(handle, torch.zeros(2, 10, device=DEV), torch.tensor([1, 10], device=DEV), 1, 1)
    assert torch.isnan(loss).item()
    # CPU tensors are refused: there is no CPU fallback
    with pytest.raises(_C.NBDTHipError):
        _C.soft_forward(handle, torch.zeros(2, 10))


def test_module_api_matches_reference_contracts(pkg_dir):
    torch.manual_seed(0)
    backbone = nn.Linear(32, 10).to(DEV)
    x = torch.randn(16, 32, device=DEV)
    y = torch.randint(0, 10, (16,), device=DEV)
    otree = O.OracleTree(*O.default_paths("CIFAR10", "induced-ResNet18", pkg_dir))

    soft = SoftNBDT("CIFAR10", backbone, arch="ResNet18")          # hierarchy = induced-ResNet18
    hard = HardNBDT("CIFAR10", backbone, hierarchy="induced-ResNet18")
    assert not soft.training and isinstance(soft.rules, SoftEmbeddedDecisionRules)
    assert isinstance(hard.rules, HardEmbeddedDecisionRules)
    assert set(soft.state_dict()) == set(backbone.state_dict())   # proxied to the backbone
    z = backbone(x)
    P = soft(x)
    assert getattr(P, "_nbdt_output_flag", False) is True and P.dtype == torch.float32
    np.testing.assert_allclose(P.detach().cpu().numpy(), O.soft_forward(otree, z.detach().cpu().numpy()),
                               rtol=2e-5, atol=1e-6)
    H = hard(x)
    assert H._nbdt_output_flag and not H.requires_grad
    assert np.array_equal(H.argmax(1).cpu().numpy(), O.hard_forward(otree, z.detach().cpu().numpy()))
    H2, decisions = hard.forward_with_decisions(x)
    assert torch.equal(H, H2) and len(decisions) == 16 and decisions[0][0]["name"] == "root"
    for dec, p in zip(decisions, H.argmax(1).tolist()):
        assert dec[-1]["node"].wnid == soft.rules.tree.wnids_leaves[p]
    # the traverse_tree classmethods (reference model.py:146, :208) on a caller-built dict == the fused kernels
    w2o = hard.rules.forward_nodes(z)
    tp, tdec = HardEmbeddedDecisionRules.traverse_tree(w2o, hard.rules.tree)
    assert torch.equal(tp, H.argmax(1)) and tp.device == z.device
    assert [[s["name"] for s in d] for d in tdec] == [[s["name"] for s in d] for d in decisions]
    assert all(abs(a["prob"] - b["prob"]) < 1e-6 for da, db in zip(tdec, decisions) for a, b in zip(da, db))
    tP = SoftEmbeddedDecisionRules.traverse_tree(soft.rules.forward_nodes(z), soft.rules.tree)
    np.testing.assert_allclose(tP.detach().cpu().numpy(), P.detach().cpu().numpy(), rtol=2e-5, atol=1e-6)
    Ps, sdec = soft.forward_with_decisions(x)
    assert len(sdec) == 16 and abs(np.prod([s["prob"] for s in sdec[3]]) - Ps[3].max().item()) < 1e-5
    with pytest.raises(NotImplementedError):
        SoftNBDT("CIFAR10", "ResNet18", arch="ResNet18")
    with pytest.raises(UserWarning):
        SoftNBDT("CIFAR10", backbone, pretrained=True, hierarchy="induced")

    # losses: fused path == composed path == oracle; NBDT outputs are rejected
    crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-ResNet18",
                           tree_supervision_weight=2.0)
    z1 = z.detach().clone().requires_grad_(True)
    l1 = crit(z1, y)
    l1.backward()
    lo, dzo = O.soft_tree_sup_loss(otree, z.detach().cpu().numpy(), y.cpu().numpy(), 1.0, 2.0)
    assert abs(l1.item() - lo) <= 1e-5 * abs(lo)
    np.testing.assert_allclose(z1.grad.cpu().numpy(), dzo, atol=1e-6, rtol=0)

    class MyCE(nn.CrossEntropyLoss):  # not `type(...) is CrossEntropyLoss` -> composed autograd path
        pass

    crit2 = SoftTreeSupLoss(dataset="CIFAR10", criterion=MyCE(), hierarchy="induced-ResNet18",
                            tree_supervision_weight=2.0)
    z2 = z.detach().clone().requires_grad_(True)
    l2 = crit2(z2, y)
    l2.backward()
    assert abs(l2.item() - lo) <= 1e-5 * abs(lo)
    np.testing.assert_allclose(z2.grad.cpu().numpy(), dzo, atol=2e-6, rtol=0)
    with pytest.raises(AssertionError):
        crit(P, y)
    # HardTreeSupLoss: fused == composed (any criterion, pooled by child count like the reference)
    hcrit = HardTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-ResNet18",
                            tree_supervision_weight=2.0)
    hcrit2 = HardTreeSupLoss(dataset="CIFAR10", criterion=MyCE(), hierarchy="induced-ResNet18",
                             tree_supervision_weight=2.0)
    hlo, hdzo = O.hard_tree_sup_loss(otree, z.detach().cpu().numpy(), y.cpu().numpy(), 1.0, 2.0)
    for c in (hcrit, hcrit2):
        zh = z.detach().clone().requires_grad_(True)
        lh = c(zh, y)
        lh.backward()
        assert abs(lh.item() - hlo) <= 1e-5 * abs(hlo)
        np.testing.assert_allclose(zh.grad.cpu().numpy(), hdzo, atol=2e-6, rtol=1e-5)
    with pytest.raises(AssertionError):
        hcrit(P, y)
    node = hcrit.tree.inodes[3]
    sel, sub, tsub = HardEmbeddedDecisionRules.get_node_logits_filtered(node, z.detach(), y.tolist())
    assert sub.shape == (sum(sel), node.num_classes) and len(tsub) == sum(sel)
    # epoch-dependent weights (nbdt/loss.py:187-189, 205-207)
    crit3 = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-ResNet18",
                            tree_supervision_weight=1.0, tree_supervision_weight_end=5.0)
    crit3.set_epoch(5, 10)
    l3 = crit3(z.detach(), y)
    lo3, _ = O.soft_tree_sup_loss(otree, z.detach().cpu().numpy(), y.cpu().numpy(), 1.0, 3.0)
    assert abs(l3.item() - lo3) <= 1e-5 * abs(lo3)


def test_soft_tree_loss_reinduces_the_hierarchy(tmp_path, pkg_dir):
    """SoftTreeLoss (reference nbdt/loss.py:269-315): cross entropy before `tree_start_epochs`, then the soft
    tree loss on a hierarchy induced from the network's own classifier rows."""
    torch.manual_seed(5)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.linear = nn.Linear(32, 10)

    net = Net()
    crit = SoftTreeLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-ResNet18", net=net,
                        arch="ResNet18", checkpoint_path=str(tmp_path / "ckpt-x.pth"), tree_start_epochs=2,
                        tree_update_every_epochs=2, tree_update_end_epochs=5, tree_supervision_weight=3.0)
    z = torch.randn(32, 10, device=DEV)
    y = torch.randint(0, 10, (32,), device=DEV)
    crit.set_epoch(1, 10)                              # before the start epoch: (w_x + w_t) * CE, weights at progress .1
    w_x, w_t = crit.current_weights()
    ce = nn.functional.cross_entropy(z, y).item()
    assert abs(crit(z, y).item() - (w_x + w_t) * ce) < 1e-5 * ce * (w_x + w_t)
    l_fast, gz = crit.loss_and_grad(z, y)
    assert abs(l_fast.item() - (w_x + w_t) * ce) < 1e-5 * ce * (w_x + w_t)
    before = [n.wnid for n in crit.tree.inodes]
    crit.set_epoch(2, 10)                              # start epoch: hierarchy re-induced from net.linear.weight
    assert os.path.exists(tmp_path / "ckpt-x" / "graph-epoch2.json")
    assert [n.wnid for n in crit.tree.inodes] != before
    otree = O.OracleTree(str(tmp_path / "ckpt-x" / "graph-epoch2.json"), os.path.join(pkg_dir, "wnids", "CIFAR10.txt"))
    lo, dzo = O.soft_tree_sup_loss(otree, z.cpu().numpy(), y.cpu().numpy(), 1.0, 3.0)
    zz = z.clone().requires_grad_(True)
    l = crit(zz, y)
    l.backward()
    assert abs(l.item() - lo) <= 1e-5 * abs(lo)
    np.testing.assert_allclose(zz.grad.cpu().numpy(), dzo, atol=1e-6, rtol=0)
    crit.set_epoch(3, 10)                              # not an update epoch: same hierarchy
    assert not os.path.exists(tmp_path / "ckpt-x" / "graph-epoch3.json")


def test_shape_and_target_contracts(pkg_dir):
    """Logits must have exactly the hierarchy's class count (wider ones used to be truncated silently); floating
    point (probability) targets never reach the fused class-index kernels: nn.CrossEntropyLoss semantics through
    the composed path instead."""
    from nbdt import _C
    from nbdt.loss import SoftTreeSupLoss
    from nbdt.model import SoftEmbeddedDecisionRules
    from nbdt.tree import Tree
    tree = Tree("CIFAR10", hierarchy="induced-wrn28_10_cifar10")
    rules = SoftEmbeddedDecisionRules(tree=tree)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(6, 10, generator=g).cuda()
    for bad in (torch.randn(6, 12).cuda(), torch.randn(6, 9).cuda()):
        with pytest.raises(_C.NBDTHipError, match="classes"):
            rules(bad)
    crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), tree=tree)
    y = torch.randint(0, 10, (6,), generator=g).cuda()
    hard_loss = crit(z, y)
    onehot = torch.nn.functional.one_hot(y, 10).float()
    soft_loss = crit(z, onehot)              # probability targets: composed path, same value for one-hot rows
    assert abs(hard_loss.item() - soft_loss.item()) < 1e-5 * abs(hard_loss.item())
    with pytest.raises(_C.NBDTHipError, match="class-index"):
        crit.loss_and_grad(z, onehot)


# ------------------------------------------------------------------------------------------------------------
# N1: classifier head + rules + SoftTreeSupLoss, forward and backward, in one launch (nbdt_head_soft_tree_loss)

@pytest.mark.parametrize("tag,K,B", [("cifar10_wrn", 640, 512), ("cifar10_r18", 512, 7), ("cifar100_wrn", 640, 256),
                                     ("tiny_r18", 512, 130), ("cifar100_wordnet", 640, 33)])
def test_fused_head_equals_linear_then_loss_then_linear_backward(tag, K, B, golden_dir, pkg_dir):
    """The fused head against the three launches it replaces (nbdt_linear_fwd -> nbdt_soft_tree_loss ->
    nbdt_linear_bwd; reference nn.Linear + nbdt/loss.py:191-203, 264-266) on the same features / weights, and against
    the numpy oracle on the logits it reports:
      * logits: bit-identical to nbdt_linear_fwd for heads below 64 classes (same one-wave-per-output arithmetic;
        wider heads take nbdt_linear_fwd's tiled GEMM: 1e-5 of the logit scale);
      * loss 1e-6 rel and dL/dz (through dL/dpooled, dL/dW, dL/db) against the unfused launches;
      * loss / dL/dz of the numpy oracle on the reported logits: 1e-5 rel / 1e-6 abs, hard decisions bit-exact;
      * ragged batches (7, 33, 130: the last block holds fewer samples than it has groups)."""
    from nbdt import ops
    _, tree, otree, ds, h = _case(tag, golden_dir, pkg_dir)
    handle = tree.device_handle(0)
    C = tree.flat.num_classes
    g = torch.Generator().manual_seed(B + K)
    pooled = torch.rand(B, K, generator=g).mul_(2.0).to(DEV)             # post-ReLU averages: non-negative
    W = torch.randn(C, K, generator=g).mul_(K ** -0.5).to(DEV)
    bias = torch.randn(C, generator=g).mul_(0.1).to(DEV)
    y = torch.randint(0, C, (B,), generator=g).to(DEV)
    for wx, wt in ((1.0, 1.0), (0.5, 10.0)):
        gW, gb = torch.zeros_like(W), torch.zeros_like(bias)
        loss, gp, z = _C.head_soft_tree_loss(handle, pooled, W, bias, y, wx, wt, gW=gW, gb=gb, want_logits=True)
        # --- the unfused sequence
        z_ref = torch.empty(B, C, device=DEV)
        ops.linear_fwd(pooled, W, bias, z_ref)
        loss_ref, gz_ref = _C.soft_tree_loss(handle, z_ref, y, wx, wt)
        gp_ref, gW_ref, gb_ref = torch.empty_like(pooled), torch.zeros_like(W), torch.zeros_like(bias)
        ops.linear_bwd(pooled, W, gz_ref, gp_ref, gW_ref, gb_ref)
        scale = z_ref.abs().max().item()
        if C < 64:
            assert torch.equal(z, z_ref)
            assert torch.equal(gp, gp_ref)          # same gz, same ascending-class fused multiply-add chain
            assert loss.item() == loss_ref.item()
        else:
            assert (z - z_ref).abs().max().item() < 1e-5 * scale
            assert (gp - gp_ref).abs().max().item() < 1e-5 * gp_ref.abs().max().item() + 1e-9
            assert abs(loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item())
        assert (gW - gW_ref).abs().max().item() < 2e-5 * gW_ref.abs().max().item() + 1e-9
        assert (gb - gb_ref).abs().max().item() < 2e-5 * gb_ref.abs().max().item() + 1e-9
        # --- the oracle on the logits the head reports
        zn = z.cpu().numpy()
        lo, dzo = O.soft_tree_sup_loss(otree, zn, y.cpu().numpy(), wx, wt)
        assert abs(loss.item() - lo) < 1e-5 * abs(lo)
        gp_oracle = torch.from_numpy(dzo).to(DEV) @ W                       # dL/dpooled = dL/dz W
        assert (gp - gp_oracle).abs().max().item() < 1e-5 * gp_oracle.abs().max().item() + 1e-8
        pred = _C.hard_forward(handle, z, want_onehot=False)[0].cpu().numpy()
        assert np.array_equal(pred, O.hard_forward(otree, zn))
    # accumulate semantics and the optional outputs
    gW2 = gW.clone()
    loss2, gp2, z2 = _C.head_soft_tree_loss(handle, pooled, W, bias, y, 0.5, 10.0, gW=gW2, want_gpooled=False)
    assert gp2 is None and z2 is None and loss2.item() == loss.item()
    assert (gW2 - 2 * gW).abs().max().item() < 1e-5 * gW.abs().max().item() + 1e-9


def test_fused_head_refuses_wide_classifiers_and_bad_shapes(golden_dir, pkg_dir):
    _, tree, _, _, _ = _case("imagenet_eff", golden_dir, pkg_dir)          # 1000 classes: the unfused path's job
    handle = tree.device_handle(0)
    pooled, W = torch.rand(4, 1280, device=DEV), torch.randn(1000, 1280, device=DEV)
    y = torch.zeros(4, dtype=torch.long, device=DEV)
    with pytest.raises(_C.NBDTHipError, match="too wide"):
        _C.head_soft_tree_loss(handle, pooled, W, None, y, 1.0, 1.0)
    crit = SoftTreeSupLoss(dataset="Imagenet1000", criterion=nn.CrossEntropyLoss(), hierarchy="induced-efficientnet_b7b")
    assert not crit.can_fuse_head(1000)
    _, tree10, _, _, _ = _case("cifar10_wrn", golden_dir, pkg_dir)
    with pytest.raises(_C.NBDTHipError, match="classifier weight"):
        _C.head_soft_tree_loss(tree10.device_handle(0), torch.rand(4, 640, device=DEV), torch.randn(12, 640, device=DEV),
                               None, y, 1.0, 1.0)
    crit10 = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
    assert crit10.can_fuse_head(10) and not crit10.can_fuse_head(100)
    assert not SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(label_smoothing=0.1),
                               hierarchy="induced-wrn28_10_cifar10").can_fuse_head(10)
