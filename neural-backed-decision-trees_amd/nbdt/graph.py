"""Induced hierarchies: agglomerative clustering of the classifier's weight rows -> graph JSON.

Restates the induced-hierarchy part of the reference's ``nbdt/graph.py``:
``MODEL_FC_KEYS`` / ``get_centers_from_*`` (:386-399, :466-511), ``build_induced_graph`` (:402-464),
``generate_graph_fname`` / ``get_graph_path_from_args`` (:194-281) and ``write_graph``
(nbdt/thirdparty/nx.py:63-66, networkx node-link JSON).  This is host-side, once-per-hierarchy work
(ward linkage on a [C, F] matrix); it produces the on-disk format ``nbdt.tree.Tree`` and the HIP
kernels consume, so new backbones / class sets are not limited to the shipped JSON files.

Differences from the reference, both forced by the target image:
* scikit-learn >= 1.4 renamed ``AgglomerativeClustering(affinity=)`` to ``metric=``; the reference's call
  (:437-439) raises ``TypeError`` there.  Same algorithm, new keyword.
* WordNet (nltk) is not installed, so inner nodes get the reference's own fallback identity -- a
  ``FakeSynset`` id ``f<number of nodes so far, 8 digits>`` with label ``(generated)``
  (nbdt/graph.py:610-615, nbdt/thirdparty/wn.py:74-94) -- instead of a common WordNet hypernym.  Node
  names are cosmetic; the tree structure and child order (what the rules layer computes with) are equal.
"""
import json
import os
from pathlib import Path

import torch

from nbdt.utils import fwd

MODEL_FC_KEYS = (
    "fc.weight", "linear.weight", "module.linear.weight", "module.net.linear.weight", "output.weight",
    "module.output.weight", "output.fc.weight", "module.output.fc.weight", "classifier.weight",
    "model.last_layer.3.weight",
)


class Graph:
    """Minimal ordered digraph with the two things the file format needs: node and edge insertion order."""

    def __init__(self):
        self.nodes = {}      # id -> {"label": ...}
        self.edges = []      # (source, target)

    def add_node(self, wnid, **attrs):
        self.nodes.setdefault(wnid, {}).update(attrs)

    def add_edge(self, source, target):
        self.add_node(source)
        self.add_node(target)
        self.edges.append((source, target))

    def succ(self, wnid):
        return [t for s, t in self.edges if s == wnid]

    def roots(self):
        targets = {t for _, t in self.edges}
        return [n for n in self.nodes if n not in targets]

    def node_link_data(self):
        nodes = [dict(attrs, id=wnid) for wnid, attrs in self.nodes.items()]
        links = [{"source": s, "target": t} for s, t in self.edges]
        return {"directed": True, "multigraph": False, "graph": {}, "nodes": nodes, "links": links}


def write_graph(G, path):
    path = str(path)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = f"{path}.tmp{os.getpid()}"          # readers never see a partial file
    with open(tmp, "w") as f:
        json.dump(G.node_link_data(), f)
    os.replace(tmp, path)


def get_centers_from_state_dict(state_dict):
    for key in MODEL_FC_KEYS:
        if key in state_dict:
            return state_dict[key].squeeze().detach()
    return None


def get_centers_from_checkpoint(checkpoint):
    data = torch.load(checkpoint, map_location=torch.device("cpu"))
    state_dict = data
    for key in ("net", "state_dict"):
        if isinstance(data, dict) and key in data:
            state_dict = data[key]
            break
    fc = get_centers_from_state_dict(state_dict)
    assert fc is not None, f"Could not find FC weights in checkpoint {checkpoint} with keys: {list(state_dict)[:8]}"
    return fc


def build_induced_graph(wnids, checkpoint=None, model=None, linkage="ward", affinity="euclidean",
                        branching_factor=2, dataset="CIFAR10", state_dict=None):
    """reference nbdt/graph.py:402-464.  `model`: an nn.Module (its state_dict is used); the reference's
    form "architecture name -> download the pretrained model" needs network access and raises here."""
    from sklearn.cluster import AgglomerativeClustering

    num_classes = len(wnids)
    assert checkpoint or model is not None or state_dict, \
        "Need to specify either `checkpoint` or `method` or `state_dict`."
    if state_dict:
        centers = get_centers_from_state_dict(state_dict)
    elif checkpoint:
        centers = get_centers_from_checkpoint(checkpoint)
    else:
        if isinstance(model, str):
            raise NotImplementedError(
                f"inducing from the pretrained `{model}` downloads a checkpoint; pass `checkpoint=` or a module")
        centers = get_centers_from_state_dict(model.state_dict())
    assert centers is not None, "Could not find FC weights (looked for: %s)" % ", ".join(MODEL_FC_KEYS)
    assert num_classes == centers.size(0), (
        f"The model FC supports {centers.size(0)} classes. However, the dataset {dataset} features "
        f"{num_classes} classes. Try passing the `--dataset` with the right number of classes.")
    centers = centers.float().cpu().numpy()

    G = Graph()
    for wnid in wnids:                       # leaves first, in class order
        G.add_node(wnid)
    clustering = AgglomerativeClustering(linkage=linkage, n_clusters=branching_factor, metric=affinity).fit(centers)
    index_to_wnid = {}
    for index, pair in enumerate(map(tuple, clustering.children_)):
        child_wnids = [wnids[c] if c < num_classes else index_to_wnid[c - num_classes] for c in pair]
        parent_wnid = "f{:08d}".format(len(G.nodes))     # FakeSynset.create_from_offset(len(G.nodes))
        G.add_node(parent_wnid, label="(generated)")
        index_to_wnid[index] = parent_wnid
        for child_wnid in child_wnids:
            G.add_edge(parent_wnid, child_wnid)
    assert len(G.roots()) == 1, G.roots()
    return G


def _checkpoint_tag(checkpoint):
    """`ckpt-<dataset>-<rest>[-induced]` -> `<rest>`; any other file name -> its stem."""
    stem = Path(checkpoint).stem
    parts = stem.split("-")
    if parts[0] == "ckpt" and len(parts) >= 3:
        return "-".join(parts[2:]).replace("-induced", "")
    return stem


def generate_graph_fname(method, seed=0, branching_factor=2, extra=0, no_prune=False, fname="", path="",
                         multi_path=False, induced_linkage="ward", induced_affinity="euclidean", checkpoint=None,
                         arch=None, **kwargs):
    """File stem of a hierarchy JSON.  The naming scheme is an on-disk contract with the reference
    (nbdt/graph.py:194-245): `graph-<method>` followed by one `-<tag><value>` suffix per non-default option,
    in a fixed order, so hierarchies written by either implementation are found by the other."""
    if path:
        return Path(path).stem
    if fname:
        return fname
    induced = method == "induced"
    if induced:
        assert checkpoint or arch, "Induced hierarchy needs either `arch` or `checkpoint`"
    suffixes = [
        (method == "random" and seed != 0, f"seed{seed}"),
        (induced and induced_linkage not in ("ward", None), f"linkage{induced_linkage}"),
        (induced and induced_affinity not in ("euclidean", None), f"affinity{induced_affinity}"),
        (induced, _checkpoint_tag(checkpoint) if checkpoint else arch),
        (method in ("random", "induced") and branching_factor != 2, f"branch{branching_factor}"),
        (extra > 0, f"extra{extra}"),
        (bool(no_prune), "noprune"),
        (bool(multi_path), "multi"),
    ]
    return "-".join([f"graph-{method}"] + [str(tag) for on, tag in suffixes if on])


def get_directory(dataset, root=None):
    return os.path.join(root or fwd(), "hierarchies", dataset)


def get_graph_path_from_args(dataset, method, path="", **kwargs):
    """reference nbdt/graph.py:248-281."""
    if path:
        return path
    return os.path.join(get_directory(dataset), generate_graph_fname(method=method, **kwargs) + ".json")
