import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch
from nbdt import ops
DEV='cuda:0'
for (B,H,cin,cout) in [(2,32,32,160),(9,8,160,320)]:
    g=torch.Generator().manual_seed(1)
    x=ops.padded(B,H,H,cin,DEV); ops.interior(x).copy_(torch.randn(B,H,H,cin,generator=g).to(torch.bfloat16))
    gy=ops.padded(B,H,H,cout,DEV); ops.interior(gy).copy_(torch.randn(B,H,H,cout,generator=g).to(torch.bfloat16))
    outs={}
    for v in (2,5):
        d=ops.conv_wgrad_desc(B,H,H,cin,cout,3,1); d.variant=v
        dw=torch.zeros(cout,9,cin,device=DEV)
        ops.conv_wgrad(d,x,gy,dw); torch.cuda.synchronize()
        outs[v]=dw.cpu()
    a,b=outs[2],outs[5]
    print(f"B={B} H={H} {cin}->{cout}: kernel {ops.last_wgrad_kernel()} nan count {torch.isnan(b).sum().item()} of {b.numel()}")
    for t in range(9):
        e=(b[:,t]-a[:,t]); nn=torch.isnan(b[:,t])
        print(f"  tap {t}: nan {nn.sum().item():6d}  max|err| (non-nan) {e[~nn].abs().max().item() if (~nn).any() else -1:.4g}  ref max {a[:,t].abs().max().item():.3g}",
              " nan by co-half:", [nn[:cout//2 if cout==160 else 80].sum().item()], " ci<16:", nn[:,:16].sum().item(), " ci>=16:", nn[:,16:32].sum().item())
