"""a captured hipGraph of BACKWARD ONLY (no RCCL, no calibration inside) vs eager launching.
WRN-28-10, 512 images, CU sharing forced; forward + fused head + SGD stay eager in both arms."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn as nn
from nbdt import engine as E, ops
from nbdt.loss import SoftTreeSupLoss
dev = torch.device("cuda:0")
crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
g = torch.Generator().manual_seed(0)
img = torch.randn(512, 3, 32, 32, generator=g).to(dev); y = torch.randint(0, 10, (512,), generator=g).to(dev)
eng = E.WRNEngine(num_classes=10, blocks=28, width_factor=10, device=dev, seed=0)
eng.set_cu_share(47.0, calibrate=False)
st = eng.store
for _ in range(4): E.train_step(eng, crit, img, y, 0.01)
torch.cuda.synchronize()

def eager_step():
    eng.zero_grad()
    pooled = eng.forward(img, training=True, head=False)
    loss, gpool, _ = crit.head_loss_and_grad(pooled, st.p("output.weight"), st.p("output.bias"), y, grad_weight=st.g("output.weight"), grad_bias=st.g("output.bias"))
    eng.backward(None, gpooled=gpool)
    eng.sgd_step(0.01, zero_grad=True)

gpool_static = torch.zeros(512, eng.feat_c, device=dev)
s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(s):          # the library's per-stream workspaces must exist on the capture stream
    for _ in range(2):
        eng.zero_grad(); pooled = eng.forward(img, training=True, head=False)
        loss, gpool, _ = crit.head_loss_and_grad(pooled, st.p("output.weight"), st.p("output.bias"), y, grad_weight=st.g("output.weight"), grad_bias=st.g("output.bias"))
        gpool_static.copy_(gpool); eng.backward(None, gpooled=gpool_static); eng.sgd_step(0.01, zero_grad=True)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s):
    eng.backward(None, gpooled=gpool_static)
    eng.join_side_stream()

def graphed_step():
    eng.zero_grad()
    pooled = eng.forward(img, training=True, head=False)
    loss, gpool, _ = crit.head_loss_and_grad(pooled, st.p("output.weight"), st.p("output.bias"), y, grad_weight=st.g("output.weight"), grad_bias=st.g("output.bias"))
    gpool_static.copy_(gpool)
    graph.replay()
    eng._grad_is_zero = False
    eng.sgd_step(0.01, zero_grad=True)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rnd in range(3):
    print(f"round {rnd}: eager {timeit(eager_step):.3f} ms/step   backward as a hipGraph {timeit(graphed_step):.3f} ms/step", flush=True)
