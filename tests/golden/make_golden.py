#!/usr/bin/env python3
"""Generate golden vectors by running the UNMODIFIED reference on CPU.

Run in the build container only (needs /root/reference):

    PYTHONHASHSEED=0 python tests/golden/make_golden.py

The reference imports torchvision / pytorchcv / nltk at module import time but
never touches them on the rules/loss path, so they are
pre-seeded in ``sys.modules`` with inert stub modules.  Nothing from the
reference is copied: the script only *calls* it and records inputs/outputs as
small ``.npz`` fixtures that travel to the GPU box (where /root/reference does
not exist).

Recorded per case (dataset, hierarchy, B, seed):
  z            [B,C] fp32 logits (torch.manual_seed(seed); randn * scale)
  y            [B]   int64 labels
  soft_P       SoftEmbeddedDecisionRules.forward(z)        nbdt/model.py:268-273
  hard_pred    HardEmbeddedDecisionRules preds (argmax of one-hot) nbdt/model.py:145-203
  loss, dz     SoftTreeSupLoss(CE)(z,y) and autograd dL/dz nbdt/loss.py:191-203,260-266
  loss_w, dz_w same with tree_supervision_weight=10, xent_weight=0.5
  hloss, hdz   HardTreeSupLoss(CE)(z,y) and autograd dL/dz nbdt/loss.py:212-257 (+ _w variants)
  node_*       per-inode logits/probs/preds/entropy (forward_nodes) nbdt/model.py:101-123
  tree_*       the reference Tree's index maps (inode order, child->classes) nbdt/tree.py:105-125
induced_*.npz: build_induced_graph (nbdt/graph.py:402-464) on seeded random classifier weights: node / link order
backbone_resnet18_*.npz: the reference's own ResNet18 (nbdt/models/resnet.py:171-179) built under torch.manual_seed,
  one train-mode batch: state-dict key list + per-tensor sums, logits, SoftTreeSupLoss loss, per-parameter gradient
  norms, BatchNorm running statistics after the forward -- the pin for oracle/torch_models.ResNet18
"""
import os
import sys
import types
import importlib.machinery
import warnings

if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class _Stub(types.ModuleType):
    """Inert module: any attribute is another stub / a dummy class."""

    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []
        self.__all__ = []
        self.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return type(item, (), {})


for name in [
    "torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.models",
    "pytorchcv", "pytorchcv.models", "pytorchcv.models.wrn_cifar", "pytorchcv.models.efficientnet",
    "nltk", "nltk.corpus",
]:
    sys.modules[name] = _Stub(name)

sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from nbdt.model import SoftEmbeddedDecisionRules, HardEmbeddedDecisionRules  # noqa: E402
from nbdt.loss import SoftTreeSupLoss, HardTreeSupLoss  # noqa: E402
from nbdt.tree import Tree  # noqa: E402

torch.set_num_threads(8)

CASES = [
    # (tag, dataset, hierarchy, B, seed, scale)
    ("cifar10_wrn", "CIFAR10", "induced-wrn28_10_cifar10", 64, 0, 3.0),
    ("cifar10_r18", "CIFAR10", "induced-ResNet18", 32, 1, 1.0),
    ("cifar10_wordnet", "CIFAR10", "wordnet", 32, 2, 2.0),
    ("cifar100_wrn", "CIFAR100", "induced-wrn28_10_cifar100", 32, 3, 3.0),
    ("cifar100_wordnet", "CIFAR100", "wordnet", 16, 4, 2.0),
    ("tiny_r18", "TinyImagenet200", "induced-ResNet18", 16, 5, 3.0),
    ("tiny_wordnet", "TinyImagenet200", "wordnet", 8, 6, 2.0),
    ("imagenet_eff", "Imagenet1000", "induced-efficientnet_b7b", 8, 7, 4.0),
]


def run_case(tag, dataset, hierarchy, B, seed, scale):
    tree = Tree(dataset, hierarchy=hierarchy)
    C = len(tree.classes)
    torch.manual_seed(seed)
    z = (torch.randn(B, C) * scale).float()
    y = torch.randint(0, C, (B,))
    # edge rows: all-zero logits (every node ties -> child 0), one huge logit
    z[0] = 0.0
    z[1] = 0.0
    z[1, C - 1] = 50.0

    soft = SoftEmbeddedDecisionRules(tree=tree)
    hard = HardEmbeddedDecisionRules(tree=tree)

    with torch.no_grad():
        P = soft(z)
        H = hard(z)
        hard_pred = H.max(1)[1]
        assert torch.all(H.sum(1) == 1)
        node_out = soft.forward_nodes(z)

    out = {
        "z": z.numpy(), "y": y.numpy().astype(np.int64),
        "soft_P": P.numpy(), "hard_pred": hard_pred.numpy().astype(np.int64),
    }

    for key, kw in [("", {}), ("_w", dict(tree_supervision_weight=10.0, xent_weight=0.5))]:
        crit = SoftTreeSupLoss(dataset=dataset, criterion=nn.CrossEntropyLoss(), tree=tree, **kw)
        zz = z.clone().requires_grad_(True)
        loss = crit(zz, y)
        loss.backward()
        out["loss" + key] = np.float32(loss.item())
        out["dz" + key] = zz.grad.numpy()
        # the other training loss: per-node cross entropy on the label's path (nbdt/loss.py:212-257)
        crit = HardTreeSupLoss(dataset=dataset, criterion=nn.CrossEntropyLoss(), tree=tree, **kw)
        zz = z.clone().requires_grad_(True)
        loss = crit(zz, y)
        loss.backward()
        out["hloss" + key] = np.float32(loss.item())
        out["hdz" + key] = zz.grad.numpy()

    # gradient of the bare rules layer under an arbitrary upstream gradient
    torch.manual_seed(seed + 100)
    gP = torch.randn(B, C)
    zz = z.clone().requires_grad_(True)
    soft(zz).backward(gP)
    out["gP"] = gP.numpy()
    out["dz_rules"] = zz.grad.numpy()

    # per-node outputs in inode order (sorted wnid), flattened child-major
    inodes = tree.inodes
    out["node_logits"] = np.concatenate([node_out[n.wnid]["logits"].numpy() for n in inodes], axis=1)
    out["node_probs"] = np.concatenate([node_out[n.wnid]["probs"].numpy() for n in inodes], axis=1)
    out["node_preds"] = np.stack([node_out[n.wnid]["preds"].numpy() for n in inodes], axis=1).astype(np.int64)
    out["node_entropy"] = np.stack([node_out[n.wnid]["entropy"].numpy() for n in inodes], axis=1)

    # tree structure pins
    wnids = [n.wnid for n in inodes]
    out["tree_inode_wnids"] = np.array(wnids)
    out["tree_root"] = np.array(tree.root.wnid)
    child_off = [0]
    child_wnid = []
    slot_off = [0]
    slot_cls = []
    for n in inodes:
        for k, ch in enumerate(n.children):
            child_wnid.append(ch.wnid)
            cls = sorted(n.child_index_to_class_index[k])
            slot_cls.extend(cls)
            slot_off.append(len(slot_cls))
        child_off.append(len(child_wnid))
    out["tree_child_off"] = np.array(child_off, dtype=np.int32)
    out["tree_child_wnid"] = np.array(child_wnid)
    out["tree_slot_off"] = np.array(slot_off, dtype=np.int32)
    out["tree_slot_cls"] = np.array(slot_cls, dtype=np.int32)
    out["tree_wnids_leaves"] = np.array(tree.wnids_leaves)

    # hard decisions for the first 4 samples: node index path + child index + prob + entropy
    _, decisions = hard.forward_with_decisions(z[:4])
    widx = {w: i for i, w in enumerate(wnids)}
    paths = np.full((4, 32), -1, dtype=np.int32)
    nexts = np.full((4, 32), -1, dtype=np.int32)
    probs = np.zeros((4, 32), dtype=np.float32)
    ents = np.zeros((4, 32), dtype=np.float32)
    for i, dec in enumerate(decisions):
        node = tree.root
        for j, step in enumerate(dec[1:]):
            paths[i, j] = widx[node.wnid]
            nexts[i, j] = step["next_index"]
            probs[i, j] = step["prob"]
            ents[i, j] = step["entropy"]
            node = step["node"]
    out["dec_path"] = paths
    out["dec_next"] = nexts
    out["dec_prob"] = probs
    out["dec_entropy"] = ents

    path = os.path.join(HERE, f"rules_{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"{tag}: C={C} N={len(inodes)} B={B} loss={out['loss']:.6f} -> {os.path.basename(path)} "
          f"({os.path.getsize(path)/1024:.1f} KiB)")


def run_induced(tag, dataset, F, seed):
    """Induced hierarchy from classifier weights (nbdt/graph.py:402-464) on seeded random weights.  The
    reference's call passes `affinity=` which scikit-learn >= 1.4 renamed to `metric=`: the class it
    imported is wrapped to translate the keyword, nothing else changes."""
    import nbdt.graph as RG
    from sklearn.cluster import AgglomerativeClustering as AC
    from networkx.readwrite.json_graph import node_link_data
    from nbdt.thirdparty.wn import get_wnids_from_dataset

    RG.AgglomerativeClustering = lambda affinity="euclidean", **kw: AC(metric=affinity, **kw)
    wnids = get_wnids_from_dataset(dataset)
    torch.manual_seed(seed)
    W = torch.randn(len(wnids), F)
    G = RG.build_induced_graph(wnids, checkpoint=None, state_dict={"linear.weight": W}, dataset=dataset)
    data = node_link_data(G)
    path = os.path.join(HERE, f"induced_{tag}.npz")
    np.savez_compressed(path, W=W.numpy(), node_ids=np.array([n["id"] for n in data["nodes"]]),
                        node_labels=np.array([n.get("label", "") for n in data["nodes"]]),
                        link_source=np.array([l["source"] for l in data["links"]]),
                        link_target=np.array([l["target"] for l in data["links"]]))
    print(f"induced {tag}: {len(data['nodes'])} nodes, {len(data['links'])} links -> {os.path.basename(path)}")


def run_backbone(tag, dataset, hierarchy, num_classes, size, batch, seed):
    """The reference's CIFAR ResNet18 is importable under the stubs: seed -> construct (default
    initialisers consume the generator in module order) -> one train-mode forward/backward with its own
    SoftTreeSupLoss.  Weights are re-creatable from the seed, so only digests of them are stored."""
    from nbdt.models.resnet import ResNet18
    torch.manual_seed(seed)
    net = ResNet18(num_classes=num_classes)
    net.train()
    g = torch.Generator().manual_seed(seed + 1000)
    x = torch.randn(batch, 3, size, size, generator=g)
    y = torch.randint(0, num_classes, (batch,), generator=g)
    keys = list(net.state_dict().keys())
    sums0 = np.array([float(v.double().sum()) for v in net.state_dict().values()])
    crit = SoftTreeSupLoss(dataset=dataset, criterion=nn.CrossEntropyLoss(), hierarchy=hierarchy)
    z = net(x)
    loss = crit(z, y)
    loss.backward()
    names = [n for n, _ in net.named_parameters()]
    gnorm = np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()])
    sd = net.state_dict()
    path = os.path.join(HERE, f"backbone_resnet18_{tag}.npz")
    np.savez_compressed(path, keys=np.array(keys), param_sums=sums0, x=x.numpy(), y=y.numpy(), logits=z.detach().numpy(),
                        loss=np.float64(loss.item()), grad_names=np.array(names), grad_norms=gnorm,
                        bn1_running_mean=sd["bn1.running_mean"].numpy(),
                        last_running_var=sd["layer4.1.bn2.running_var"].numpy(), seed=np.int64(seed))
    print(f"backbone resnet18 {tag}: loss {loss.item():.6f}, {len(keys)} state-dict entries -> {os.path.basename(path)}")


if __name__ == "__main__":
    run_backbone("cifar10", "CIFAR10", "induced-ResNet18", 10, 32, 8, 21)
    run_backbone("tiny200", "TinyImagenet200", "induced-ResNet18", 200, 64, 4, 22)
    for case in CASES:
        run_case(*case)
    run_induced("cifar10", "CIFAR10", 64, 11)
    run_induced("cifar100", "CIFAR100", 64, 12)
    run_induced("tiny200", "TinyImagenet200", 32, 13)
