import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E
from nbdt.loss import SoftTreeSupLoss
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(1234)
img = torch.randn(B, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (B,), generator=g).to(dev)

def grads():
    eng.zero_grad()
    z = eng.forward(img, training=True)
    loss, gz = crit.loss_and_grad(z, y)
    eng.backward(gz)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in eng.named_params("grad").items()}

def report(tag, a, b):
    rows = []
    for k in a:
        rows.append((((a[k].float() - b[k].float()).norm() / (b[k].float().norm() + 1e-30)).item(), k))
    rows.sort(reverse=True)
    print(f"== {tag}: worst tensors", flush=True)
    for r, k in rows[:6]:
        print(f"   {r:.3e} {k}")

eng.set_cu_share(None)
g0 = grads(); g1 = grads()
report("off vs off", g1, g0)
orig = E.Backbone._share_cus if hasattr(E, "Backbone") else None
cls = type(eng).__mro__[1]
real = cls._share_cus
cls._share_cus = lambda self, e, t: 0
eng.set_cu_share(47, 230, 16, 96, high_priority=False)
report("reorder only (ordinary kernels, no priority)", grads(), g0)
eng.set_cu_share(47, 230, 16, 96, high_priority=True)
report("reorder only, priority stream", grads(), g0)
cls._share_cus = real
eng.set_cu_share(47, 230, 16, 96, high_priority=False)
report("share (cus kernels), no priority", grads(), g0)
eng.set_cu_share(None)
report("off again", grads(), g0)
