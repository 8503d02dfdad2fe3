import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add(oracle=True)
import torch, torch.nn as nn, numpy as np
import nbdt_oracle as O, torch_models as TM
from nbdt import engine as E, ops
from nbdt.loss import SoftTreeSupLoss
DEV='cuda:0'
torch.manual_seed(0)
ref = TM.WRN(10, 10, 2); eng = E.WRNEngine(num_classes=10, blocks=10, width_factor=2, device=DEV, seed=0)
eng.load_state_dict(ref.state_dict())
pkg=os.path.join(nbdt_path.PKG_DIR,'nbdt')
otree = O.OracleTree(*O.default_paths("CIFAR10", "induced-wrn28_10_cifar10", pkg))
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(1)
x = torch.randn(16, 3, 32, 32, generator=g); y = torch.randint(0, 10, (16,), generator=g)
ref.train()
# capture oracle intermediates
caps = {}
def hook(name):
    def f(mod, inp, out):
        t = out[0] if isinstance(out, tuple) else out
        t.retain_grad(); caps[name] = t
    return f
for i in (1,2,3):
    u = getattr(ref.features, f'stage{i}').unit1
    u.register_forward_hook(hook(f's{i}.out'))
    u.body.conv1.register_forward_hook(hook(f's{i}.t'))      # tuple (t, a1)
    u.body.conv2.register_forward_hook(hook(f's{i}.u'))
    u.body.conv1.bn.register_forward_hook(hook(f's{i}.bn1'))
    u.body.conv2.bn.register_forward_hook(hook(f's{i}.bn2'))
with TM.emulate_bf16():
    z = ref(x)
    loss, dz = O.soft_tree_sup_loss(otree, z.detach().numpy(), y.numpy())
    z.backward(torch.from_numpy(dz))
eng.zero_grad()
ze = eng.forward(x.to(DEV), training=True)
l, gz = crit.loss_and_grad(ze, y.to(DEV))
def rel(a,b):
    a=a.float().cpu().flatten(); b=b.float().cpu().flatten(); return ((a-b).norm()/(b.norm()+1e-30)).item()
print('logits', rel(ze, z.detach()), 'gz', rel(gz, torch.from_numpy(dz)))
nhwc = lambda t: t.detach().permute(0,2,3,1)
for u in eng.units:
    k=u['key']; i=k[1]
    print(k, 'fwd out', rel(ops.interior(u['x_out']), nhwc(caps[f's{i}.out'])),
          't', rel(ops.interior(eng._bufs[[kk for kk in eng._bufs if kk[0]==k+'.t'][0]]), nhwc(caps[f's{i}.t'])))
# run backward manually with captures
orig = ops.bn_bwd
eng.backward(gz)
torch.cuda.synchronize()
# engine grads vs oracle intermediate grads: g wrt unit outputs
# reconstruct: after backward, buffers hold last-written grads; stage3 out grad is in g_out128 buffer
for kk,v in eng._bufs.items():
    if isinstance(kk[0], str) and kk[0].startswith(('g_out','g_in','ga2','gt','ga1')):
        print(kk, float(ops.interior(v).float().norm()) if v.dim()==4 else None)
go = [kk for kk in eng._bufs if kk[0].startswith('g_out')][0]
print('g wrt s3.out', rel(ops.interior(eng._bufs[go]), nhwc(caps['s3.out'].grad)))
for name,key in [('s3.u (= conv2 out grad)', None)]:
    pass
ga2 = [kk for kk in eng._bufs if kk[0]=='ga2_128'][0]
print('stage3 ga2 vs grad of bn2 out', rel(ops.interior(eng._bufs[ga2]), nhwc(caps['s3.bn2'].grad)) if caps['s3.bn2'].grad is not None else 'n/a')
gt = [kk for kk in eng._bufs if kk[0]=='gt_128'][0]
print('stage3 gt vs grad of t', rel(ops.interior(eng._bufs[gt]), nhwc(caps['s3.t'].grad)))
grads = eng.named_params('grad')
for n,p in ref.named_parameters():
    if 'stage3' in n or 'post' in n: print(n, rel(grads[n], p.grad))

print("==== self-consistency of pool_bn_bwd on the engine's own tensors")
import torch.nn.functional as F
xl = ops.interior(eng._x_last).float().cpu().permute(0,3,1,2).clone().requires_grad_(True)
pb = eng.post_bn
gam = pb.gamma.cpu().clone(); bet = pb.beta.cpu().clone()
a = F.relu(F.batch_norm(xl, None, None, gam, bet, training=True, eps=1e-5))
pooled = a.mean((2,3))
gpool = [v for kk,v in eng._bufs.items() if kk[0]=='gpool'][0].cpu()
pooled.backward(gpool)
print('engine g vs torch recompute from engine x_last/gpool:', rel(ops.interior(eng._bufs[go]), xl.grad.permute(0,2,3,1)))
mask_e = (a.detach()>0)
with torch.no_grad():
    xo = caps['s3.out'].detach()
    ao = F.relu(F.batch_norm(xo, None, None, ref.features.post_activ.bn.weight, ref.features.post_activ.bn.bias, training=True, eps=1e-5))
print('x_last rel err', rel(xl.detach(), xo), 'mask disagreement fraction', (mask_e != (ao>0)).float().mean().item())
d = (xl.detach()-xo).abs(); print('fraction of x_last elements that differ at all:', (d>0).float().mean().item(), 'max', d.max().item())
