import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch
from nbdt import ops
DEV='cuda:0'
B,H,cin,cout=2,32,32,160
x=ops.padded(B,H,H,cin,DEV); ops.interior(x).fill_(1.0)
gy=ops.padded(B,H,H,cout,DEV); ops.interior(gy).fill_(1.0)
for v in (2,5):
    d=ops.conv_wgrad_desc(B,H,H,cin,cout,3,1); d.variant=v
    dw=torch.zeros(cout,9,cin,device=DEV)
    ops.conv_wgrad(d,x,gy,dw); torch.cuda.synchronize()
    o=dw.cpu()
    print("variant",v,ops.last_wgrad_kernel(),"nan",torch.isnan(o).sum().item())
    for t in range(9):
        vals,counts=torch.unique(o[:,t],return_counts=True)
        print("  tap",t,[(float(a),int(c)) for a,c in zip(vals[:6],counts[:6])])
