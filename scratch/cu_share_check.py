import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path
nbdt_path.add()
import torch
from nbdt import ops
DEV = "cuda:0"
torch.manual_seed(0)
for (B, H, W, C) in [(64, 32, 32, 160), (8, 16, 16, 320), (3, 8, 8, 640), (2, 4, 4, 32)]:
    def act(scale=1.0):
        p = ops.padded(B, H, W, C, DEV)
        ops.interior(p).copy_((torch.randn(B, H, W, C) * scale).to(torch.bfloat16).to(DEV))
        return p
    gy, x, add = act(), act(2.0), act()
    mean, rstd = torch.randn(C, device=DEV) * 0.1, torch.rand(C, device=DEV) + 0.5
    gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV) * 0.3
    rows = (B * H * W + 255) // 256
    part = torch.randn(rows * 2 * C, device=DEV)
    for use_add in (False, True):
        outs = []
        for cus in (0, 48, 7, 256):
            dsum, dg, db = torch.empty(2 * C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            gx = ops.padded(B, H, W, C, DEV)
            ops.bn_bwd_fused(gy, x, mean, rstd, gamma, beta, part, dsum, dg, db, gx, gx_add=add if use_add else None, cus=cus)
            torch.cuda.synchronize()
            outs.append(gx.float())
        for o, cus in zip(outs[1:], (48, 7, 256)):
            print((B, H, W, C), "add" if use_add else "   ", "cus", cus, "max abs diff to ordinary:", (o - outs[0]).abs().max().item(),
                  "border zero:", o[:, 0].abs().max().item() == 0 and o[:, :, 0].abs().max().item() == 0)
# weight gradient with a CU budget
for (B, H, W, cin, cout) in [(64, 32, 32, 160, 160), (64, 16, 16, 320, 320), (64, 8, 8, 640, 640)]:
    d = ops.conv_wgrad_desc(B, H, W, cin, cout, 3, 1)
    xp, gp = ops.padded(B, H, W, cin, DEV), ops.padded(B, H, W, cout, DEV)
    ops.interior(xp).copy_(torch.randn(B, H, W, cin).to(torch.bfloat16).to(DEV))
    ops.interior(gp).copy_(torch.randn(B, H, W, cout).to(torch.bfloat16).to(DEV))
    res = []
    for budget in (0, 208, 64):
        dw = torch.zeros(cout * 9 * cin, device=DEV)
        ops.conv_wgrad(d, xp, gp, dw, cu_budget=budget)
        torch.cuda.synchronize()
        res.append(dw)
        print((B, H, W, cin, cout), "budget", budget, ops.last_wgrad_kernel(), "rel diff to budget 0:",
              ((dw - res[0]).norm() / res[0].norm()).item())
