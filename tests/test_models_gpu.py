"""Drop-in model API: nbdt.models factories + SoftNBDT/HardNBDT + SoftTreeSupLoss used exactly like
the reference's own tests (tests/test_inference.py:6-45, tests/test_train.py:9-49), plus value checks
against the fp32 CPU oracle backbones with identical weights."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import nbdt_oracle as O
import torch_models as TM

pytestmark = pytest.mark.gpu

from nbdt import engine as E  # noqa: E402
from nbdt.loss import SoftTreeSupLoss  # noqa: E402
from nbdt.model import HardNBDT, SoftNBDT  # noqa: E402
from nbdt.models import ResNet18, wrn28_10_cifar10  # noqa: E402

DEV = "cuda:0"


def _cos(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


@pytest.mark.parametrize("dataset,classes,size", [("CIFAR10", 10, 32), ("CIFAR100", 100, 32),
                                                  ("TinyImagenet200", 200, 64)])
def test_reference_style_inference_and_training(dataset, classes, size):
    # reference tests/test_inference.py: wrap ResNet18 with Soft/HardNBDT(hierarchy="induced"), one sample
    model = ResNet18(num_classes=classes)
    x = torch.randn(1, 3, size, size).to(DEV)
    y = torch.randint(0, classes, (1,)).to(DEV)
    soft = SoftNBDT(dataset=dataset, model=model, hierarchy="induced")
    hard = HardNBDT(dataset=dataset, model=model, hierarchy="induced")
    assert not model.training            # NBDT puts the wrapped model in eval mode
    with torch.no_grad():
        P = soft(x)
        H = hard(x)
    assert P.shape == (1, classes) and abs(P.sum().item() - 1) < 1e-4
    assert H.shape == (1, classes) and H.sum().item() == 1
    # reference tests/test_train.py: SoftTreeSupLoss on the backbone's logits, backward through it
    model.train()
    crit = SoftTreeSupLoss(dataset=dataset, criterion=nn.CrossEntropyLoss(), hierarchy="induced")
    loss = crit(model(x), y)
    loss.backward()
    assert torch.isfinite(loss).item()
    grads = [p.grad for p in model.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    assert sum(g.abs().sum().item() for g in grads) > 0


def test_resnet18_matches_fp32_oracle_and_state_dict_roundtrip(pkg_dir):
    torch.manual_seed(0)
    ref = TM.ResNet18(10)
    net = ResNet18(num_classes=10)
    assert set(net.state_dict()) == set(ref.state_dict())
    for k, v in ref.state_dict().items():
        assert tuple(net.state_dict()[k].shape) == tuple(v.shape), k
    net.load_state_dict(ref.state_dict())
    otree = O.OracleTree(*O.default_paths("CIFAR10", "induced-ResNet18", pkg_dir))
    crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-ResNet18")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(32, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (32,), generator=g)
    ref.train(); net.train()
    z_ref = ref(x)
    loss_ref, dz = O.soft_tree_sup_loss(otree, z_ref.detach().numpy(), y.numpy())
    z_ref.backward(torch.from_numpy(dz))
    z = net(x.to(DEV))
    loss = crit(z, y.to(DEV))
    loss.backward()
    scale = z_ref.abs().max().item()
    assert (z.detach().cpu() - z_ref.detach()).abs().max().item() < 3e-2 * scale
    assert abs(loss.item() - float(loss_ref)) < 2e-2 * abs(float(loss_ref))
    rp = dict(ref.named_parameters())
    for name, p in net.named_parameters():
        c = _cos(p.grad, rp[name].grad)
        ratio = p.grad.float().norm().item() / rp[name].grad.norm().item()
        # 17 ReLU layers: bf16 storage flips ~0.3% of the masks per layer (see test_engine_gpu.py), so
        # the deepest (stem) gradient agrees with the fp32 oracle in direction to ~0.94
        assert c > 0.90 and abs(ratio - 1) < 0.10, f"{name}: cos {c:.4f} ratio {ratio:.4f}"
    # eval-mode decisions (running stats) agree with the oracle on the same inputs
    ref.eval(); net.eval()
    with torch.no_grad():
        agree = (net(x.to(DEV)).argmax(1).cpu() == ref(x).argmax(1)).float().mean().item()
    assert agree >= 0.9
    # round trip through a fresh module
    net2 = ResNet18(num_classes=10, seed=5)
    net2.load_state_dict(net.state_dict())
    net2.eval()
    with torch.no_grad():
        assert torch.equal(net2(x.to(DEV)), net(x.to(DEV)))


def test_torch_optimizer_on_facade_equals_fused_engine_step(pkg_dir):
    """optim.SGD over the facade's parameters (views of the flat buffer) == the engine's fused SGD."""
    a = ResNet18(num_classes=10, seed=3)
    b = ResNet18(num_classes=10, seed=3)
    crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-ResNet18")
    opt = torch.optim.SGD(a.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(16, 3, 32, 32, generator=g).to(DEV)
    y = torch.randint(0, 10, (16,), generator=g).to(DEV)
    a.train(); b.train()
    for _ in range(3):
        opt.zero_grad()
        la = crit(a(x), y)
        la.backward()
        opt.step()
        lb = E.train_step(b.engine, crit, x, y, lr=0.05)
    assert abs(la.item() - lb.item()) < 2e-2 * abs(lb.item())
    fa, fb = a.engine.store.flat, b.engine.store.flat
    # two separate runs differ by fp32 atomic ordering (wgrad split-K, BN sums) amplified by bf16
    # rounding; the optimizers themselves agree to 1e-5 (test_sgd_matches_torch_optim)
    assert ((fa - fb).norm() / fb.norm()).item() < 3e-2


def test_wrn_facade_names_and_forward():
    net = wrn28_10_cifar10()
    sd = net.state_dict()
    assert "features.stage2.unit1.identity_conv.weight" in sd and "output.weight" in sd
    assert tuple(sd["features.stage1.unit1.body.conv1.conv.weight"].shape) == (160, 16, 3, 3)
    assert tuple(sd["features.init_block.weight"].shape) == (16, 3, 3, 3)
    n_conv = sum(v.numel() for k, v in sd.items() if v.dim() == 4)
    assert n_conv == 36454832
    soft = SoftNBDT(dataset="CIFAR10", model=net, arch="wrn28_10_cifar10")
    with torch.no_grad():
        P = soft(torch.randn(4, 3, 32, 32).to(DEV))
    np.testing.assert_allclose(P.sum(1).cpu().numpy(), 1.0, atol=1e-4)


@pytest.mark.parametrize("arch", ["ResNet18", "wrn", "efficientnet_b0"])
def test_fused_inference_path_equals_the_unfused_one(arch):
    """Eval-mode BatchNorm folded into the conv epilogues (engine.fuse_eval) vs separate BN passes."""
    from nbdt.engine import ResNetEngine, WRNEngine, train_step
    from nbdt.engine_effnet import EfficientNetEngine
    from nbdt.loss import SoftTreeSupLoss
    import torch.nn as nn
    if arch == "ResNet18":
        eng, ds, h, size, C = ResNetEngine(10, device=DEV, seed=1), "CIFAR10", "induced-ResNet18", 32, 10
    elif arch == "wrn":
        eng, ds, h, size, C = WRNEngine(10, blocks=10, width_factor=2, device=DEV, seed=1), "CIFAR10", "induced-wrn28_10_cifar10", 32, 10
    else:
        eng, ds, h, size, C = EfficientNetEngine(1000, device=DEV, seed=1), "Imagenet1000", "induced-efficientnet_b7b", 64, 1000
    crit = SoftTreeSupLoss(dataset=ds, criterion=nn.CrossEntropyLoss(), hierarchy=h)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16, 3, size, size, generator=g).to(DEV)
    y = torch.randint(0, C, (16,), generator=g).to(DEV)
    for _ in range(3):                                    # make the running statistics non-trivial
        train_step(eng, crit, x, y, 0.02)
    eng.fuse_eval = True
    zf = eng.forward(x, training=False).clone()
    eng.fuse_eval = False
    zu = eng.forward(x, training=False).clone()
    scale = zu.abs().max().item()
    assert (zf - zu).abs().max().item() < 3e-2 * scale, ((zf - zu).abs().max().item(), scale)
    assert (zf.argmax(1) == zu.argmax(1)).float().mean().item() >= 0.9
    # a further training step invalidates the folded transforms
    train_step(eng, crit, x, y, 0.02)
    eng.fuse_eval = True
    z2 = eng.forward(x, training=False)
    assert (z2 - zf).abs().max().item() > 0


def test_backward_refuses_stale_or_eval_mode_activations():
    """The engine keeps ONE set of activations and only the batch-statistics BatchNorm backward: a backward that
    would silently read another forward's activations, or differentiate an eval-mode forward, must raise."""
    model = ResNet18(num_classes=10)
    crit = nn.CrossEntropyLoss()
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(4, 3, 32, 32, generator=g).to(DEV), torch.randn(4, 3, 32, 32, generator=g).to(DEV)
    y = torch.randint(0, 10, (4,), generator=g).to(DEV)
    model.train()
    la = crit(model(a), y)
    lb = crit(model(b), y)                   # second forward overwrites the activations of the first
    lb.backward()                            # the most recent forward differentiates fine
    with pytest.raises(RuntimeError, match="another forward ran"):
        la.backward()
    la = crit(model(a), y)
    with torch.no_grad():
        model(b)                             # a grad-free forward overwrites them too
    with pytest.raises(RuntimeError, match="another forward ran"):
        la.backward()
    model.eval()
    le = crit(model(a), y)
    with pytest.raises(RuntimeError, match="eval-mode forward"):
        le.backward()
