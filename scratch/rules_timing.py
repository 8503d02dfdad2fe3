"""Phase stamps of soft_fwd_kernel (timing build, -DNBDT_RULES_TIMING=1): s_memtime ticks (100 MHz) per phase, per wave
of block 0.  Run with NBDT_HIP_LIB pointing at the variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbdt_path; nbdt_path.add()
import torch
from nbdt import _C
from nbdt.tree import Tree
NAMES = ["offsets+logits->LDS", "gather(slot_cls)", "slot chains", "node softmax", "gather(cls_slot)", "path products+store"]
for B, ds, h in [(512, "CIFAR10", "induced-wrn28_10_cifar10"), (1024, "TinyImagenet200", "induced-ResNet18"),
                 (256, "Imagenet1000", "induced-efficientnet_b7b")]:
    tree = Tree(ds, hierarchy=h); C = len(tree.classes); hd = tree.device_handle(0)
    z = torch.randn(B, C, device="cuda:0") * 3
    for _ in range(5): P = _C.soft_forward(hd, z)
    torch.cuda.synchronize()
    st = P.flatten()[:32].cpu().view(4, 8)
    nw = 1 if (C <= 64 and tree.flat.num_slots <= 64) else 4 if (C <= 512 and tree.flat.num_slots <= 512) else 16
    print(f"== ({B},{C}) waves/sample {nw}: s_memtime ticks since kernel start, per wave of block 0")
    for w in range(4):
        row = st[w].tolist()
        print(f"  wave {w}: " + "  ".join(f"{NAMES[i]} {row[i + 1] - row[i]:.0f}" for i in range(6)) + f"  | total {row[6]:.0f}"
              f"  (node softmax: first trip {row[7] - row[3]:.0f}, second trip {row[4] - row[7]:.0f})")
