#!/usr/bin/env python3
"""Bench configuration (WRN-28-10, 512 images) against the bf16-EMULATING oracle (torch_models.emulate_bf16: fp32
arithmetic, activations / gradients / weights rounded to bf16 at the engine's storage points): how much of the
engine-vs-fp32-oracle gradient disagreement is storage rounding the oracle can reproduce?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add(oracle=True)
import torch, torch.nn as nn
import nbdt_oracle as O, torch_models as TM
from nbdt import engine as E, ops
from nbdt.loss import SoftTreeSupLoss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
ref = TM.WRN(10, 28, 10)
otree = O.OracleTree(*O.default_paths("CIFAR10", "induced-wrn28_10_cifar10", os.path.join(nbdt_path.PKG_DIR, "nbdt")))
g = torch.Generator().manual_seed(11)
x = torch.randn(B, 3, 32, 32, generator=g); y = torch.randint(0, 10, (B,), generator=g)
sd = {k: v.clone() for k, v in ref.state_dict().items()}
ref.train()
res = {}
for tag, ctx in (("fp32", None), ("bf16-emulating", TM.emulate_bf16)):
    ref.load_state_dict(sd); ref.zero_grad()
    t0 = time.time()
    if ctx is None:
        z = ref(x)
    else:
        with ctx():
            z = ref(x)
    loss, dz = O.soft_tree_sup_loss(otree, z.detach().numpy(), y.numpy())
    if ctx is None:
        z.backward(torch.from_numpy(dz))
    else:
        with ctx():
            z.backward(torch.from_numpy(dz))
    res[tag] = (z.detach(), float(loss), {n: p.grad.clone() for n, p in ref.named_parameters()})
    print(f"{tag} oracle: {time.time() - t0:.1f} s, loss {float(loss):.5f}", flush=True)
ops.set_deterministic(True)
eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device="cuda:0", seed=0)
eng.load_state_dict(sd)
eng.set_cu_share(47.0, calibrate=False)
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
eng.zero_grad()
zz = eng.forward(x.cuda(), training=True)
l, gz = crit.loss_and_grad(zz, y.cuda())
eng.backward(gz); torch.cuda.synchronize()
grads = eng.named_params("grad")
cos = lambda a, b: (a.float().cpu().flatten() @ b.flatten() / (a.float().norm().cpu() * b.norm() + 1e-30)).item()
names = [n for n in res["fp32"][2]]
for tag in res:
    zr, lr_, gr = res[tag]
    cs = {n: cos(grads[n], gr[n]) for n in names}
    conv = [n for n in names if n.endswith("conv.weight")]
    print(f"engine vs {tag}: logit err {(zz.cpu() - zr).abs().max().item() / zr.abs().max().item():.4f} of scale, loss {l.item():.5f} vs {lr_:.5f}; "
          f"gradient cosine min {min(cs.values()):.4f} ({min(cs, key=cs.get)}), first dense conv {cs[conv[0]]:.4f}, last conv {cs[conv[-1]]:.4f}, "
          f"mean over conv weights {sum(cs[n] for n in conv) / len(conv):.4f}")
a, b = res["fp32"][2], res["bf16-emulating"][2]
cs = {n: cos(a[n], b[n]) for n in names}
print(f"fp32 oracle vs bf16-emulating oracle (no engine involved): gradient cosine min {min(cs.values()):.4f}, first dense conv "
      f"{cs[[n for n in names if n.endswith('conv.weight')][0]]:.4f}")
