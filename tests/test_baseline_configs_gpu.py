"""Backbone + rules parity at the BASELINE.json configurations other than the benched one (configs[2..4]; configs[1],
the bench line, is tests/test_engine_gpu.py::test_bench_configuration_step_matches_fp32_oracle): one training step of
the HIP engine against the fp32 CPU oracle port with IDENTICAL weights and inputs, same metrics everywhere --

  * logits: max error over the logit scale, argmax agreement;
  * loss: relative error of SoftTreeSupLoss (fused HIP kernel vs numpy oracle, each on its own logits);
  * hard decisions: the HardNBDT kernel on the ENGINE's logits must equal the numpy oracle's rules on the SAME
    logits bit for bit (the rules layer's contract), and agree with the oracle's decisions on its own logits;
  * gradients: per-parameter cosine and norm ratio against the oracle's autograd.

Tolerances are bf16 storage against fp32 arithmetic (stated per test).  Shards: config 3 is a 1024-image global batch
on 4 GPUs = 256 images per rank, configs 4 / 5 are 8-GPU data-parallel with 128 / 8 images per rank here; a rank's step
is what a 1-GPU box can check (tests/test_dist*.py cover the exchange)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import nbdt_oracle as O
import torch_models as TM

pytestmark = pytest.mark.gpu

from nbdt import _C  # noqa: E402
from nbdt import engine as E  # noqa: E402
from nbdt.loss import SoftTreeSupLoss  # noqa: E402
from nbdt.tree import Tree  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def deterministic_mode():
    """These tests assert tolerances on ONE step at random initialisation, where the order of fp32 atomics alone moves
    small BatchNorm-parameter gradients by several percent from run to run.  With
    nbdt_set_deterministic the engine's step is a pure function of weights and inputs, so a threshold that holds once
    holds on every box."""
    from nbdt import ops
    ops.set_deterministic(True)
    yield
    ops.set_deterministic(False)


def _cos(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def _train_step_vs_oracle(ref, eng, dataset, hierarchy, pkg_dir, x, y, w_t=1.0):
    """One train-mode forward + SoftTreeSupLoss + backward through both paths; returns the comparison metrics."""
    otree = O.OracleTree(*O.default_paths(dataset, hierarchy, pkg_dir))
    crit = SoftTreeSupLoss(dataset=dataset, criterion=nn.CrossEntropyLoss(), hierarchy=hierarchy,
                           tree_supervision_weight=w_t)
    ref.train()
    z_ref = ref(x)
    loss_ref, dz = O.soft_tree_sup_loss(otree, z_ref.detach().numpy(), y.numpy(), 1.0, w_t)
    z_ref.backward(torch.from_numpy(dz))
    z_ref = z_ref.detach()

    eng.zero_grad()
    z = eng.forward(x.to(DEV), training=True)
    loss, gz = crit.loss_and_grad(z, y.to(DEV))
    eng.backward(gz)
    torch.cuda.synchronize()

    tree = Tree(dataset, hierarchy=hierarchy)
    zc = z.float().cpu()
    hard = _C.hard_forward(tree.device_handle(0), z.float(), want_onehot=False)[0].cpu().numpy()
    m = {
        "scale": z_ref.abs().max().item(),
        "logit_err": (zc - z_ref).abs().max().item(),
        "argmax": (zc.argmax(1) == z_ref.argmax(1)).float().mean().item(),
        "loss": loss.item(), "loss_ref": float(loss_ref),
        # rules contract: HIP kernel == numpy oracle on the SAME logits, bit for bit
        "hard_same_logits": bool(np.array_equal(hard, O.hard_forward(otree, zc.numpy()))),
        "hard_vs_ref": float((hard == O.hard_forward(otree, z_ref.numpy())).mean()),
        "soft_argmax_same_logits": bool(np.array_equal(
            _C.soft_forward(tree.device_handle(0), z.float()).argmax(1).cpu().numpy(),
            O.soft_forward(otree, zc.numpy()).argmax(1))),
    }
    grads = eng.named_params("grad")
    rows = []
    for name, p in ref.named_parameters():
        gn = p.grad.norm().item()
        rows.append((name, _cos(grads[name], p.grad), grads[name].float().norm().item() / (gn + 1e-30), gn))
    m["grads"] = rows
    return m


def _report(tag, m):
    print(f"[{tag}] logit err {m['logit_err']:.4g} of scale {m['scale']:.4g}; loss {m['loss']:.5f} vs "
          f"{m['loss_ref']:.5f}; argmax {m['argmax']:.4f}; hard (same logits) {m['hard_same_logits']}, "
          f"hard vs oracle's own {m['hard_vs_ref']:.4f}")
    live = [r for r in m["grads"] if r[3] > 1e-6]
    worst = min(live, key=lambda r: r[1])
    print(f"[{tag}] worst gradient cosine {worst[1]:.4f} ({worst[0]}); conv/linear weight norm ratios "
          f"{min(r[2] for r in live if r[0].endswith('weight')):.4f}..{max(r[2] for r in live if r[0].endswith('weight')):.4f}")


def test_config3_wrn28_10_cifar100_rank_shard(pkg_dir):
    """configs[2]: WideResNet28x10 + SoftTreeSupLoss on CIFAR100 (100-leaf induced-wrn28_10_cifar100 hierarchy),
    global batch 1024 on 4 GPUs -> this rank's 256 images.  Architecture: reference nbdt/models/wideresnet.py:1-5
    (pytorchcv wrn28_10_cifar100)."""
    torch.manual_seed(0)
    ref = TM.WRN(100, 28, 10)
    eng = E.WRNEngine(num_classes=100, blocks=28, width_factor=10, device=DEV, seed=0)
    eng.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(31)
    x = torch.randn(256, 3, 32, 32, generator=g)
    y = torch.randint(0, 100, (256,), generator=g)
    m = _train_step_vs_oracle(ref, eng, "CIFAR100", "induced-wrn28_10_cifar100", pkg_dir, x, y)
    _report("config 3", m)
    assert m["logit_err"] < 3e-2 * m["scale"]
    assert abs(m["loss"] - m["loss_ref"]) < 2e-2 * abs(m["loss_ref"])
    assert m["hard_same_logits"] and m["soft_argmax_same_logits"]
    assert m["argmax"] >= 0.95 and m["hard_vs_ref"] >= 0.93
    for name, c, ratio, gn in m["grads"]:
        if name.endswith("conv.weight"):
            assert abs(ratio - 1) < 0.03, (name, ratio)
        assert c > 0.85, (name, c)                      # 25 ReLU layers of 1-ulp bf16 mask flips (see config 2's test)


def test_config4_resnet18_tinyimagenet200_hard_nbdt(pkg_dir):
    """configs[3]: ResNet18 + HardNBDT on TinyImagenet200 (200 leaves, 64x64 images, induced-ResNet18 hierarchy):
    a SoftTreeSupLoss training step with tree-supervision weight 10 (reference scripts/gen_train_eval_wideresnet.sh:4)
    at 128 images per rank, then eval-mode HardNBDT predictions (reference nbdt/model.py:145-203) from the updated
    running statistics.  The oracle backbone is pinned to the reference's own class
    (tests/golden/backbone_resnet18_tiny200.npz, reference nbdt/models/resnet.py:171-179)."""
    torch.manual_seed(0)
    ref = TM.ResNet18(200)
    eng = E.ResNetEngine(num_classes=200, device=DEV, seed=0)
    eng.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(41)
    x = torch.randn(128, 3, 64, 64, generator=g)
    y = torch.randint(0, 200, (128,), generator=g)
    m = _train_step_vs_oracle(ref, eng, "TinyImagenet200", "induced-ResNet18", pkg_dir, x, y, w_t=10.0)
    _report("config 4", m)
    assert m["logit_err"] < 3e-2 * m["scale"]
    assert abs(m["loss"] - m["loss_ref"]) < 2e-2 * abs(m["loss_ref"])
    assert m["hard_same_logits"] and m["soft_argmax_same_logits"]
    assert m["argmax"] >= 0.95 and m["hard_vs_ref"] >= 0.93
    for name, c, ratio, gn in m["grads"]:
        # conv / linear weights carry the step; the 64- to 512-element BatchNorm vectors are noisier (measured: weight
        # norm ratios 0.959..1.068, BatchNorm-parameter ratios up to 1.10, worst cosine 0.927)
        tol = 0.10 if (name.endswith("conv1.weight") or name.endswith("conv2.weight") or "linear" in name
                       or name.endswith("shortcut.0.weight")) else 0.20
        assert c > 0.90 and abs(ratio - 1) < tol, (name, c, ratio)
    # ---- HardNBDT inference through the drop-in module (eval mode: running statistics of the step above)
    from nbdt.model import HardNBDT
    from nbdt.models import ResNet18
    net = ResNet18(num_classes=200)
    sd = {k: v for k, v in eng.state_dict().items()}
    net.load_state_dict(sd)
    hard = HardNBDT(dataset="TinyImagenet200", model=net, hierarchy="induced-ResNet18")
    with torch.no_grad():
        onehot = hard(x.to(DEV))
        z_eval = net(x.to(DEV)).float().cpu()
    pred = onehot.argmax(1).cpu().numpy()
    otree = O.OracleTree(*O.default_paths("TinyImagenet200", "induced-ResNet18", pkg_dir))
    assert torch.equal(onehot.sum(1).cpu(), torch.ones(128)) and getattr(onehot, "_nbdt_output_flag", False)
    assert np.array_equal(pred, O.hard_forward(otree, z_eval.numpy()))      # bit-exact on the same logits
    ref.eval()                       # the oracle's running statistics moved in its own training forward above
    with torch.no_grad():
        z_ref_eval = ref(x)
    assert (z_eval - z_ref_eval).abs().max().item() < 4e-2 * z_ref_eval.abs().max().item()
    agree = float((pred == O.hard_forward(otree, z_ref_eval.numpy())).mean())
    print(f"[config 4] eval-mode HardNBDT agreement with the oracle's own decisions: {agree:.4f}")
    assert agree >= 0.90


def test_config5_efficientnet_b0_imagenet1000(pkg_dir):
    """configs[4]: EfficientNet-B0 + SoftNBDT on the 1000-leaf Imagenet1000 induced-efficientnet_b7b hierarchy,
    224x224 images, 32 images of a rank's shard; dropout off so that both paths are deterministic functions of the same
    weights.  Two comparators: the fp32 CPU oracle, and the SAME oracle rounding to bf16 at the engine's storage points
    (torch_models.emulate_bf16: raw conv outputs, stored activations, 1x1-conv weights, the matching gradients) --
    against the latter the logits must agree to 5 % of their scale (~80 storage points between image and logits, each
    renormalised by a BatchNorm; what is left is summation order snapping to 1-ulp bf16 differences)."""
    from nbdt.engine_effnet import EfficientNetEngine
    torch.manual_seed(0)
    ref = TM.EfficientNetB0(num_classes=1000, dropout_rate=0.0)
    init = {k: v.clone() for k, v in ref.state_dict().items()}
    eng = EfficientNetEngine(num_classes=1000, dropout_rate=0.0, device=DEV)
    eng.load_state_dict(init)
    g = torch.Generator().manual_seed(51)
    x = torch.randn(32, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (32,), generator=g)
    m = _train_step_vs_oracle(ref, eng, "Imagenet1000", "induced-efficientnet_b7b", pkg_dir, x, y)
    _report("config 5 vs fp32 oracle", m)
    assert m["logit_err"] < 0.06 * m["scale"]       # measured 3.1 % (the 8-image version of round 3 allowed 15 %)
    assert abs(m["loss"] - m["loss_ref"]) < 2e-2 * abs(m["loss_ref"])
    assert m["hard_same_logits"] and m["soft_argmax_same_logits"]
    bad = []
    for name, c, ratio, gn in m["grads"]:
        if gn < 1e-6:
            continue        # mathematically zero gradients (a BatchNorm shift feeding conv -> BatchNorm)
        if not (c > 0.93 and abs(ratio - 1) < 0.25):     # measured: worst cosine 0.98, norm ratios 0.87 .. 1.06
            bad.append((name, round(c, 4), round(ratio, 4)))
    assert not bad, bad
    # second comparator: the bf16-emulating oracle, same weights, same batch (one more engine step on fresh statistics)
    ref2 = TM.EfficientNetB0(num_classes=1000, dropout_rate=0.0)
    ref2.load_state_dict(init)
    eng.load_state_dict(init)
    with TM.emulate_bf16():
        m2 = _train_step_vs_oracle(ref2, eng, "Imagenet1000", "induced-efficientnet_b7b", pkg_dir, x, y)
    _report("config 5 vs bf16-emulating oracle", m2)
    assert m2["logit_err"] < 0.05 * m2["scale"], (m2["logit_err"], m2["scale"])
    assert abs(m2["loss"] - m2["loss_ref"]) < 1e-2 * abs(m2["loss_ref"])
    # (same exclusion as above, by NAME: under emulation the rounding noise gives the mathematically zero gradients a
    # small random value)
    zero = {r[0] for r in m["grads"] if r[3] < 1e-6}
    live = [r for r in m2["grads"] if r[0] not in zero]
    assert min(r[1] for r in live) > 0.88, min(live, key=lambda r: r[1])
    # SoftNBDT inference output of the same logits: probabilities, rows sum to 1, equal to the oracle's on those logits
    tree = Tree("Imagenet1000", hierarchy="induced-efficientnet_b7b")
    z = eng.forward(x.to(DEV), training=False).float()
    P = _C.soft_forward(tree.device_handle(0), z).cpu().numpy()
    otree = O.OracleTree(*O.default_paths("Imagenet1000", "induced-efficientnet_b7b", pkg_dir))
    np.testing.assert_allclose(P, O.soft_forward(otree, z.cpu().numpy()), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(P.sum(1), 1.0, atol=1e-5)
