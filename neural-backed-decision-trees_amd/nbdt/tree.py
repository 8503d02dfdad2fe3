"""Hierarchy data model: ``Tree`` / ``Node`` plus the flat CSR form the HIP kernels consume.

Same public surface as the reference's ``nbdt/tree.py`` (Tree :145-241, Node
:38-142) for the members the rules layer and losses use: ``tree.inodes`` (sorted
by wnid), ``tree.root``, ``tree.classes``, ``tree.wnids_leaves``,
``tree.wnid_to_node``, ``tree.wnid_to_class_index``; ``node.children``,
``node.num_classes``, ``node.child_index_to_class_index``,
``node.class_index_to_child_index``, ``node.is_leaf()``.

The graph file is networkx node-link JSON (``nodes[{id,label}]``,
``links[{source,target}]``); it is parsed directly (no networkx needed): node
iteration order = ``nodes`` order, child order = order of first appearance in
``links`` -- what ``node_link_graph`` + ``G.succ`` yield (reference
nbdt/thirdparty/nx.py:69-75).  Leaf order inside a child's class list is
ascending class index (the reference's order comes from iterating a Python
``set`` and is not reproducible; only the sum order of a mean depends on it).
"""
import json
import os
from collections import defaultdict

import numpy as np

from nbdt.utils import (
    DATASETS,
    DATASET_TO_NUM_CLASSES,
    dataset_to_default_path_graph,
    dataset_to_default_path_wnids,
    hierarchy_to_path_graph,
)


def dataset_to_dummy_classes(dataset):
    """reference nbdt/tree.py:20-23 (FakeSynset wnids 'f%08d')."""
    assert dataset in DATASETS, f"unknown dataset {dataset}"
    return ["f{:08d}".format(i) for i in range(DATASET_TO_NUM_CLASSES[dataset])]


def read_graph(path):
    """node-link JSON -> (node ids in file order, succ, pred, labels)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"No such hierarchy file: {path}")
    with open(path) as f:
        g = json.load(f)
    order, labels = [], {}
    succ, pred = {}, {}

    def touch(w):
        if w not in succ:
            succ[w] = []
            pred[w] = []
            order.append(w)

    for n in g["nodes"]:
        touch(n["id"])
        labels[n["id"]] = n.get("label", "(generated)")
    for e in g.get("links", g.get("edges", [])):
        s, t = e["source"], e["target"]
        touch(s)
        touch(t)
        if t not in succ[s]:
            succ[s].append(t)
            pred[t].append(s)
    return order, succ, pred, labels


def get_wnids(path_wnids):
    """reference nbdt/thirdparty/wn.py:24-31."""
    if not os.path.exists(path_wnids):
        raise FileNotFoundError(f"No such wnids file: {path_wnids}")
    with open(path_wnids) as f:
        return [w.strip() for w in f.readlines()]


class Node:
    """One hierarchy node (reference nbdt/tree.py:38-142)."""

    def __init__(self, tree, wnid):
        self.tree = tree
        self.wnid = wnid
        self.name = tree._labels.get(wnid, "(generated)")
        self.original_classes = tree.classes
        self.num_original_classes = len(tree.wnids_leaves)
        self.num_children = len(self.succ)
        self.num_classes = self.num_children
        self.class_index_to_child_index, self.child_index_to_class_index = self.build_class_mappings()
        self.classes = [
            ",".join(str(self.original_classes[o]) for o in olds)
            for _, olds in sorted(self.child_index_to_class_index.items())
        ]
        self.leaves = sorted(tree._leaves_under(wnid))
        self.num_leaves = len(self.leaves)

    @property
    def succ(self):
        return self.tree._succ[self.wnid]

    @property
    def pred(self):
        return self.tree._pred[self.wnid]

    @property
    def children(self):
        return [self.tree.wnid_to_node[w] for w in self.succ]

    @property
    def parents(self):
        return [self.tree.wnid_to_node[w] for w in self.pred]

    @property
    def parent(self):
        return self.parents[0] if self.parents else None

    def is_leaf(self):
        return len(self.succ) == 0

    def is_root(self):
        return len(self.pred) == 0

    # thin delegates a reference subclass may call (reference nbdt/tree.py:96-97, 127-139); the kernels never do
    def get_leaves(self):
        """wnids of the leaves under this node, in the Tree's own order."""
        under = self.tree._leaves_under(self.wnid)
        return [w for w in self.tree._order if w in under]

    def build_classes(self):
        return list(self.classes)

    @property
    def class_counts(self):
        """Number of original classes under each child."""
        return [len(olds) for _, olds in sorted(self.child_index_to_class_index.items())]

    @staticmethod
    def dim(nodes):
        return sum(node.num_classes for node in nodes)

    def wnid_to_class_index(self, wnid):
        return self.tree.wnid_to_class_index[wnid]

    def wnid_to_child_index(self, wnid):
        return list(self.succ).index(wnid)

    def build_class_mappings(self):
        """reference nbdt/tree.py:105-125 (no `other` class: never enabled by Tree)."""
        if self.is_leaf():
            return {}, {}
        old_to_new = defaultdict(list)
        new_to_old = defaultdict(list)
        for new_index, child in enumerate(self.succ):
            leaves = self.tree._leaves_under(child)
            for old_index in sorted(self.tree.wnid_to_class_index[l] for l in leaves
                                    if l in self.tree.wnid_to_class_index):
                old_to_new[old_index].append(new_index)
                new_to_old[new_index].append(old_index)
        return old_to_new, new_to_old


class FlatTree:
    """CSR form of a Tree for the kernels (see include/nbdt_hip.h: nbdt_tree_create)."""

    def __init__(self, tree):
        inodes = tree.inodes
        index = {n.wnid: i for i, n in enumerate(inodes)}
        self.num_classes = C = len(tree.classes)
        self.num_inodes = len(inodes)
        self.root = index[tree.root.wnid]
        node_off, slot_off, slot_cls, slot_next = [0], [0], [], []
        per_class = [[] for _ in range(C)]
        self.multi_path_node = None      # first inner node with a class under two of its children (DAG hierarchies)
        for n in inodes:
            under = [c for k in range(len(n.children)) for c in n.child_index_to_class_index[k]]
            if self.multi_path_node is None and len(set(under)) != len(under):
                self.multi_path_node = n.wnid
            for k, child in enumerate(n.children):
                slot = len(slot_next)
                cls = list(n.child_index_to_class_index[k])
                if not cls:
                    raise ValueError(f"child {child.wnid} of {n.wnid} has no leaf in the class list")
                if len(set(cls)) != len(cls):
                    raise AssertionError("All old indices must be unique")  # model.py:237-240
                slot_cls.extend(cls)
                slot_off.append(len(slot_cls))
                for c in cls:
                    per_class[c].append(slot)
                if child.is_leaf():
                    slot_next.append(-tree.wnid_to_class_index[child.wnid] - 1)
                else:
                    slot_next.append(index[child.wnid])
            node_off.append(len(slot_next))
        cls_off, cls_slot = [0], []
        for c in range(C):
            cls_slot.extend(per_class[c])
            cls_off.append(len(cls_slot))
        i32 = lambda a: np.ascontiguousarray(np.array(a, dtype=np.int32))
        self.node_off, self.slot_off, self.slot_cls = i32(node_off), i32(slot_off), i32(slot_cls)
        self.cls_off, self.cls_slot, self.slot_next = i32(cls_off), i32(cls_slot), i32(slot_next)
        self.num_slots = len(slot_next)
        self.inode_wnids = [n.wnid for n in inodes]


    def require_single_path(self):
        """The soft rules multiply one child probability per (node, class); a class under two children of one
        node has no single factor.  The reference asserts the same thing per call (nbdt/model.py:237-240)."""
        if self.multi_path_node is not None:
            raise AssertionError("All old indices must be unique in order for this operation to be correct "
                                 f"(node {self.multi_path_node} reaches a class through two children)")


class Tree:
    """reference nbdt/tree.py:145-241."""

    def __init__(self, dataset, path_graph=None, path_wnids=None, classes=None, hierarchy=None):
        if dataset and hierarchy and not path_graph:
            path_graph = hierarchy_to_path_graph(dataset, hierarchy)
        if dataset and not path_graph:
            path_graph = dataset_to_default_path_graph(dataset)
        if dataset and not path_wnids:
            path_wnids = dataset_to_default_path_wnids(dataset)
        if dataset and not classes:
            classes = dataset_to_dummy_classes(dataset)
        self.load_hierarchy(dataset, path_graph, path_wnids, classes)

    def load_hierarchy(self, dataset, path_graph, path_wnids, classes):
        self.dataset = dataset
        self.path_graph = path_graph
        self.path_wnids = path_wnids
        self.classes = classes
        self._order, self._succ, self._pred, self._labels = read_graph(path_graph)
        self.wnids_leaves = get_wnids(path_wnids)
        self.wnid_to_class = {w: c for w, c in zip(self.wnids_leaves, self.classes)}
        self.wnid_to_class_index = {w: i for i, w in enumerate(self.wnids_leaves)}
        self._leaf_memo = {}
        self.wnid_to_node = {}
        for w in self._order:
            self.wnid_to_node[w] = Node(self, w)
        self.nodes = [self.wnid_to_node[w] for w in sorted(self.wnid_to_node)]
        self.inodes = [n for n in self.nodes if not n.is_leaf()]
        self.leaves = [self.wnid_to_node[w] for w in self.wnids_leaves]
        self._flat = None
        self._handles = {}

    def _leaves_under(self, wnid):
        memo = self._leaf_memo
        if wnid in memo:
            return memo[wnid]
        stack = [wnid]
        # iterative post-order so 18-deep ImageNet trees and wide DAGs do not recurse
        out = set()
        seen = set()
        while stack:
            w = stack.pop()
            if w in seen:
                continue
            seen.add(w)
            if w in memo:
                out |= memo[w]
            elif not self._succ[w]:
                out.add(w)
            else:
                stack.extend(self._succ[w])
        memo[wnid] = frozenset(out)
        return memo[wnid]

    def update_from_model(self, model, arch, dataset, classes=None, path_wnids=None, path_graph=None):
        """reference nbdt/tree.py:176-190: re-induce the hierarchy from `model`'s classifier weights, write
        it to `path_graph` and reload this Tree in place (kernel handles are rebuilt lazily)."""
        from nbdt.hierarchy import generate_hierarchy
        assert model is not None, "`model` cannot be NoneType"
        path_graph = generate_hierarchy(dataset=dataset, method="induced", arch=arch, model=model, path=path_graph,
                                        path_wnids=path_wnids or self.path_wnids)
        tree = Tree(dataset, path_graph=path_graph, path_wnids=path_wnids or self.path_wnids,
                    classes=classes or self.classes, hierarchy="induced")
        self.load_hierarchy(dataset=tree.dataset, path_graph=tree.path_graph, path_wnids=tree.path_wnids,
                            classes=tree.classes)

    @classmethod
    def create_from_args(cls, args, classes=None):
        return cls(args.dataset, args.path_graph, args.path_wnids, classes=classes,
                   hierarchy=args.hierarchy)

    @property
    def root(self):
        for node in self.inodes:
            if node.is_root():
                return node
        raise UserWarning("Should not be reachable. Tree should always have root")

    def get_wnid_to_node(self):
        """reference nbdt/tree.py:209-213: a fresh {wnid: Node} map (load_hierarchy keeps its own in wnid_to_node)."""
        return {w: Node(self, w) for w in self._order}

    @property
    def flat(self):
        if self._flat is None:
            self._flat = FlatTree(self)
        return self._flat

    def device_handle(self, device_index):
        """Per-device immutable kernel handle (created lazily, cached)."""
        from nbdt import _C
        h = self._handles.get(device_index)
        if h is None:
            h = _C.TreeHandle(self.flat, device_index)
            self._handles[device_index] = h
        return h

    def get_leaf_to_steps(self):
        """reference nbdt/tree.py:215-229 (BFS root->leaf paths with child indices)."""
        leaf_to_path = {}
        roots = [w for w in self._order if not self._pred[w]]
        for root in roots:
            frontier = [(root, 0, [])]
            while frontier:
                node, child_index, path = frontier.pop(0)
                path = path + [(child_index, node)]
                if not self._succ[node]:
                    leaf_to_path[node] = path
                    continue
                frontier.extend((c, i, path) for i, c in enumerate(self._succ[node]))
        leaf_to_steps = {}
        for leaf in self.wnids_leaves:
            nxt = [i for i, _ in leaf_to_path[leaf][1:]] + [-1]
            leaf_to_steps[leaf] = [
                {"node": self.wnid_to_node[w], "name": self.wnid_to_node[w].name, "next_index": ni}
                for ni, (_, w) in zip(nxt, leaf_to_path[leaf])
            ]
        return leaf_to_steps
