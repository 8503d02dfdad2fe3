"""CPU ORACLE for the NBDT rules layer + SoftTreeSupLoss -- TEST INFRASTRUCTURE ONLY.

This file is a plain numpy restatement of the reference's algorithm.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; the product package (``neural-backed-decision-trees_amd/nbdt``)
never does and fails loudly when the HIP library is missing.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py``
against ``tests/golden/rules_*.npz`` -- outputs of the UNMODIFIED reference run
in the build container by ``tests/golden/make_golden.py`` (the reference's own
tests hold no golden vectors).  Parity is therefore pinned
against the reference itself, not against this restatement alone.

Reference lines restated (paths relative to /root/reference):
  hierarchy load ............ nbdt/tree.py:160-174, nbdt/thirdparty/nx.py:69-75,
                              nbdt/thirdparty/wn.py:24-31
  class<->child maps ........ nbdt/tree.py:105-125  (Node.build_class_mappings)
  node logits (mean) ........ nbdt/model.py:83-99   (get_node_logits)
  preds/probs/entropy ....... nbdt/model.py:101-120 (get_all_node_outputs)
  soft path product ......... nbdt/model.py:207-242 (SoftEmbeddedDecisionRules.traverse_tree)
  hard traversal ............ nbdt/model.py:145-192 (HardEmbeddedDecisionRules.traverse_tree)
  tree-supervision loss ..... nbdt/loss.py:187-203, 260-266

Arithmetic conventions (these define "bit-exact" for decision indices):
  * all arithmetic in IEEE fp32;
  * a child's logit is the SEQUENTIAL fp32 sum of its leaves' logits in
    ascending class-index order, then one fp32 division by the leaf count
    (torch CPU ``mean`` = ``sum().div_(n)``; the reference's own leaf order
    comes from iterating a Python set and is not reproducible);
  * argmax takes the FIRST maximum (torch.max(dim=1) on CPU);
  * the soft product multiplies 1.0 by the node probabilities in inode order
    (nodes sorted by wnid), exactly like ``class_probs[:, old] *= probs[:, new]``.
"""
import json
import os

import numpy as np

F32 = np.float32
EPS32 = np.finfo(np.float32).eps


class OracleTree:
    """Index maps of one hierarchy (restates Tree/Node, nbdt/tree.py:38-174)."""

    def __init__(self, path_graph, path_wnids):
        with open(path_graph) as f:
            g = json.load(f)
        with open(path_wnids) as f:
            self.wnids_leaves = [w.strip() for w in f.readlines()]
        self.num_classes = len(self.wnids_leaves)
        cls_index = {w: i for i, w in enumerate(self.wnids_leaves)}

        succ = {n["id"]: [] for n in g["nodes"]}
        pred = {n["id"]: [] for n in g["nodes"]}
        for e in g["links"]:
            succ.setdefault(e["source"], [])
            succ.setdefault(e["target"], [])
            pred.setdefault(e["source"], [])
            pred.setdefault(e["target"], [])
            if e["target"] not in succ[e["source"]]:
                succ[e["source"]].append(e["target"])
                pred[e["target"]].append(e["source"])
        self.succ, self.pred = succ, pred

        memo = {}

        def leaves_under(w):
            # get_leaves(G, child): leaves among descendants(child) | {child}
            key = w
            if key in memo:
                return memo[key]
            if not succ[w]:
                out = {w}
            else:
                out = set()
                for c in succ[w]:
                    out |= leaves_under(c)
            memo[key] = out
            return out

        self.inode_wnids = sorted(w for w in succ if succ[w])  # tree.py:172-173
        self.inode_index = {w: i for i, w in enumerate(self.inode_wnids)}
        roots = [w for w in self.inode_wnids if not pred[w]]
        self.root = self.inode_index[roots[0]]  # tree.py:202-207

        # per inode: list over children of sorted class-index lists
        self.children = []      # [n] -> [child wnid]
        self.child_classes = []  # [n][k] -> [class idx ascending]
        for w in self.inode_wnids:
            self.children.append(list(succ[w]))
            per_child = []
            for c in succ[w]:
                per_child.append(sorted(cls_index[l] for l in leaves_under(c) if l in cls_index))
            self.child_classes.append(per_child)

    @property
    def num_inodes(self):
        return len(self.inode_wnids)


def default_paths(dataset, hierarchy, root_dir):
    """hierarchy name -> files (nbdt/utils.py:62-71)."""
    return (os.path.join(root_dir, "hierarchies", dataset, f"graph-{hierarchy}.json"),
            os.path.join(root_dir, "wnids", f"{dataset}.txt"))


# --------------------------------------------------------------------------- rules

def node_logits(tree, z):
    """nbdt/model.py:83-99 -- list over inodes of [B, K_n] fp32 child logits."""
    z = np.asarray(z, dtype=F32)
    out = []
    for per_child in tree.child_classes:
        cols = []
        for cls in per_child:
            acc = np.zeros(z.shape[0], dtype=F32)
            for c in cls:                      # sequential fp32 sum, ascending class order
                acc = (acc + z[:, c]).astype(F32)
            cols.append((acc / F32(len(cls))).astype(F32))
        out.append(np.stack(cols, axis=1))
    return out


def _softmax(s):
    m = s.max(axis=1, keepdims=True)
    e = np.exp((s - m).astype(F32)).astype(F32)
    return (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)


def _entropy(p):
    # torch.distributions.Categorical(probs=p).entropy(): probs renormalised,
    # logits = log(clamp(probs, eps, 1-eps)), H = -sum(p * logits)
    p = (p / p.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
    logit = np.log(np.clip(p, EPS32, F32(1.0) - EPS32)).astype(F32)
    return (-(p * logit).sum(axis=1, dtype=F32)).astype(F32)


def node_outputs(tree, z):
    """nbdt/model.py:101-120 -- per inode dict(logits, preds, probs, entropy)."""
    outs = []
    for s in node_logits(tree, z):
        p = _softmax(s)
        outs.append({"logits": s, "preds": np.argmax(s, axis=1).astype(np.int64),
                     "probs": p, "entropy": _entropy(p)})
    return outs


def soft_forward(tree, z, outs=None):
    """nbdt/model.py:207-242, 268-273 -- [B,C] fp32 path probabilities."""
    outs = outs or node_outputs(tree, z)
    B = np.asarray(z).shape[0]
    P = np.ones((B, tree.num_classes), dtype=F32)
    for n, o in enumerate(outs):
        for k, cls in enumerate(tree.child_classes[n]):
            if cls:
                P[:, cls] = (P[:, cls] * o["probs"][:, k:k + 1]).astype(F32)
    return P


def hard_forward(tree, z, outs=None, with_decisions=False):
    """nbdt/model.py:145-192 -- [B] int64 predicted class (greedy root->leaf walk)."""
    outs = outs or node_outputs(tree, z)
    B = np.asarray(z).shape[0]
    cls_index = {w: i for i, w in enumerate(tree.wnids_leaves)}
    pred = np.zeros(B, dtype=np.int64)
    decisions = []
    for b in range(B):
        n = tree.root
        steps = []
        while True:
            k = int(outs[n]["preds"][b])
            steps.append((n, k, float(outs[n]["probs"][b, k]), float(outs[n]["entropy"][b])))
            child = tree.children[n][k]
            if child in tree.inode_index:
                n = tree.inode_index[child]
            else:
                pred[b] = cls_index[child]
                break
        decisions.append(steps)
    return (pred, decisions) if with_decisions else pred


def hard_onehot(tree, pred):
    """nbdt/model.py:188-192 -- rows of eye(C)."""
    return np.eye(tree.num_classes, dtype=F32)[pred]


# --------------------------------------------------------------------------- loss

def _cross_entropy_rows(x, y):
    """nn.CrossEntropyLoss per-row terms and softmax (rows treated as logits)."""
    m = x.max(axis=1, keepdims=True)
    e = np.exp((x - m).astype(F32)).astype(F32)
    ssum = e.sum(axis=1, keepdims=True, dtype=F32)
    lse = (np.log(ssum).astype(F32) + m).astype(F32)[:, 0]
    rows = (lse - x[np.arange(x.shape[0]), y]).astype(F32)
    return rows, (e / ssum).astype(F32)


def tree_weight(progress, start, end, power=1.0):
    """nbdt/loss.py:187-189."""
    p = progress ** power
    return (1 - p) * start + p * end


def soft_tree_sup_loss(tree, z, y, w_xent=1.0, w_tree=1.0):
    """nbdt/loss.py:191-203, 260-266: w_x*CE(z,y) + w_t*CE(P,y) (P fed to CE as logits).

    Returns (loss fp32 scalar, dL/dz [B,C] fp32) -- the gradient is the closed
    form of SURVEY Appendix B, checked against the reference's autograd goldens.
    """
    z = np.asarray(z, dtype=F32)
    y = np.asarray(y, dtype=np.int64)
    B, C = z.shape
    outs = node_outputs(tree, z)
    P = soft_forward(tree, z, outs)
    rx, sx = _cross_entropy_rows(z, y)
    rt, st = _cross_entropy_rows(P, y)
    loss = F32(w_xent) * rx.mean(dtype=F32) + F32(w_tree) * rt.mean(dtype=F32)

    onehot = np.zeros((B, C), dtype=F32)
    onehot[np.arange(B), y] = 1
    dz = ((sx - onehot) * F32(w_xent / B)).astype(F32)
    g = ((st - onehot) * F32(w_tree / B)).astype(F32)          # dL/dP
    dz += rules_backward(tree, z, g, outs=outs, P=P)
    return F32(loss), dz.astype(F32)


def hard_tree_sup_loss(tree, z, y, w_xent=1.0, tree_supervision_weight=1.0, w_tree=None):
    """nbdt/loss.py:191-203 + 212-257 (HardTreeSupLoss) with criterion = CrossEntropyLoss.

    For every inner node the samples whose label lies under the node form that node's
    training set (model.py:127-143, get_node_logits_filtered: child index of the label =
    ``class_index_to_child_index[y][0]``); rows are pooled by the node's number of children K and
    each pool contributes ``CE_mean(pool) * len(pool) / (B*N/2) * tree_supervision_weight``
    (loss.py:228, 250-256).  ``TreeSupLoss.forward`` then multiplies the sum by the scheduled
    tree weight AGAIN (:195-203) -- ``w_tree`` here, which defaults to the attribute like the
    reference's end==start schedule.  Returns (loss, dL/dz) with the closed-form gradient.
    """
    z = np.asarray(z, dtype=F32)
    y = np.asarray(y, dtype=np.int64)
    B, C = z.shape
    w_tree = tree_supervision_weight if w_tree is None else w_tree
    logits = node_logits(tree, z)
    num_losses = B * tree.num_inodes / 2.0
    rx, sx = _cross_entropy_rows(z, y)
    onehot = np.zeros((B, C), dtype=F32)
    onehot[np.arange(B), y] = 1
    dz = ((sx - onehot) * F32(w_xent / B)).astype(F32)

    pools = {}  # K -> (list of [rows,K] logits, list of targets, list of (node, rows))
    for n, per_child in enumerate(tree.child_classes):
        child_of = {}
        for k, cls in enumerate(per_child):
            for c in cls:
                child_of.setdefault(c, k)          # cls[0]: first child holding the label
        rows = np.array([b for b in range(B) if int(y[b]) in child_of], dtype=np.int64)
        K = len(per_child)
        pool = pools.setdefault(K, ([], [], []))
        pool[0].append(logits[n][rows])
        pool[1].extend(child_of[int(y[b])] for b in rows)
        pool[2].append((n, rows))
    loss_tree = F32(0)
    for K, (subs, tgts, where) in pools.items():
        sub = np.concatenate(subs, axis=0)
        if not sub.shape[0]:
            continue
        tg = np.array(tgts, dtype=np.int64)
        rows_ce, sm = _cross_entropy_rows(sub, tg)
        fraction = sub.shape[0] / float(num_losses) * tree_supervision_weight
        loss_tree = F32(loss_tree + rows_ce.mean(dtype=F32) * F32(fraction))
        hot = np.zeros_like(sm)
        hot[np.arange(sm.shape[0]), tg] = 1
        ds_all = ((sm - hot) * F32(w_tree * tree_supervision_weight / num_losses)).astype(F32)
        o = 0
        for n, rows in where:
            ds = ds_all[o:o + len(rows)]
            o += len(rows)
            for k, cls in enumerate(tree.child_classes[n]):
                for c in cls:
                    dz[rows, c] += (ds[:, k] / F32(len(cls))).astype(F32)
    loss = F32(w_xent) * rx.mean(dtype=F32) + F32(w_tree) * loss_tree
    return F32(loss), dz.astype(F32)


def rules_backward(tree, z, gP, outs=None, P=None):
    """dL/dz of the soft rules layer for an upstream gradient gP = dL/dP."""
    z = np.asarray(z, dtype=F32)
    outs = outs or node_outputs(tree, z)
    P = soft_forward(tree, z, outs) if P is None else P
    Pg = (P * gP).astype(F32)
    dz = np.zeros_like(z)
    for n, o in enumerate(outs):
        G = np.stack([Pg[:, cls].sum(axis=1, dtype=F32) for cls in tree.child_classes[n]], axis=1)
        ds = (G - o["probs"] * G.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
        for k, cls in enumerate(tree.child_classes[n]):
            dz[:, cls] += (ds[:, k:k + 1] / F32(len(cls))).astype(F32)
    return dz
