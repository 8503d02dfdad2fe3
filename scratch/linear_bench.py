"""nbdt_linear_fwd / nbdt_linear_bwd (gemm_f32_kernel for >= 64 classes) at the EfficientNet-B0 head's shape."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "neural-backed-decision-trees_amd"))
import torch
from nbdt import ops

DEV = "cuda:0"


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, K, N in ((128, 1280, 1000), (32, 1280, 1000), (128, 512, 200), (256, 640, 100)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.03).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    gz = torch.randn(B, N, generator=g).to(DEV)
    z, gx = torch.empty(B, N, device=DEV), torch.empty(B, K, device=DEV)
    gw, gb = torch.zeros_like(w), torch.zeros_like(b)
    ops.linear_fwd(x, w, b, z)
    ops.linear_bwd(x, w, gz, gx, gw, gb)
    zr = x.double() @ w.double().t() + b.double()
    gxr, gwr = gz.double() @ w.double(), gz.double().t() @ x.double()
    rel = lambda a, r: ((a.double() - r).norm() / r.norm()).item()
    errs = f"z {rel(z, zr):.1e} gx {rel(gx, gxr):.1e} gw {rel(gw, gwr):.1e} gb {rel(gb, gz.double().sum(0)):.1e}"
    t_f = timed(lambda: ops.linear_fwd(x, w, b, z))
    t_x = timed(lambda: ops.linear_bwd(x, w, gz, gx, None, None))
    t_w = timed(lambda: ops.linear_bwd(x, w, gz, None, gw, gb))
    print(f"B={B} K={K} N={N}: fwd {t_f:.1f} us  dgrad {t_x:.1f} us  wgrad+bias {t_w:.1f} us   rel err vs fp64: "
          f"{errs}", flush=True)
