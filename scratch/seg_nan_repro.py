import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch, torch.nn.functional as F
from nbdt import ops
from nbdt._C import lib, check, ptr
from nbdt.ops import stream_ptr
DEV = "cuda:0"
sys.path.insert(0, os.path.join(nbdt_path.ROOT, "tests"))
import test_conv_seg_gpu as T
B, Hi, Wi, cin, cout, tile, nbuf = 5, 8, 8, 96, 160, 256, 3
xf, xp = T._rand_act(B, Hi, Wi, cin, seed=1)
w_oihw, wb = T._rand_weight(cout, cin, 3, seed=2)
Ho, Wo = Hi // 2, Wi // 2
plan = ops.seg_fwd_s2(B, Hi, Wi, cin, cout, tile=tile, nbuf=nbuf)
xs = T._s2d(xp, B, Hi, Wi, cin)
wt = plan.tile_weights([wb])
# poison LDS with NaN patterns: a wgrad / other kernels leave arbitrary bits there
poison = torch.full((1 << 20,), float("nan"), device=DEV)
sink = torch.zeros(4, device=DEV)
for trial in range(6):
    if trial % 2 == 1:
        # fp32 NaN-heavy kernel traffic through LDS: the BN statistics kernels use LDS for their folds
        big = ops.padded(64, 32, 32, 160, DEV); ops.interior(big).fill_(float("nan"))
        scr2 = torch.zeros(ops.BN_SLOTS * 2 * 2048, device=DEV)
        m, r = torch.empty(160, device=DEV), torch.empty(160, device=DEV)
        ops.bn_stats(big, scr2, m, r)
    out2 = ops.padded(B, Ho, Wo, cout, DEV)
    M = B * Ho * Wo
    scr = torch.full((((M + 255) // 256) * 2 * cout,), 7.0, device=DEV)
    plan([xs], wt, out2, bn_scratch=scr)
    torch.cuda.synchronize()
    bad = torch.isnan(scr).sum().item()
    print(trial, "nan in partials:", bad, "nan in out:", torch.isnan(out2.float()).sum().item(), scr[:4].tolist())
