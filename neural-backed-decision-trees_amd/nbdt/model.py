"""NBDT inference wrappers and decision rules, MI355X-native.

Drop-in surface of the reference's ``nbdt/model.py``: ``EmbeddedDecisionRules``,
``HardEmbeddedDecisionRules``, ``SoftEmbeddedDecisionRules`` (reference :65-273) and ``NBDT``,
``HardNBDT``, ``SoftNBDT`` (:281-373) with the same constructor signatures, methods and
behavioural contracts (``_nbdt_output_flag``, eval-on-init, state_dict proxying, error types).
Every forward dispatches to one fused HIP kernel in libnbdt_hip.so (csrc/rules.hip) instead of
the reference's per-node Python loop; there is no CPU path.

Out of scope here: the segmentation variants (SegNBDT*) and pretrained
checkpoint download (`pretrained=True` needs network access the target box does not have).
"""
import torch
import torch.nn as nn

from nbdt import _C
from nbdt.tree import Tree

model_urls = {}  # reference :27-57 lists release URLs; no network on the target -> not mirrored


class _SoftRulesFn(torch.autograd.Function):
    """P = soft_rules(z); backward recomputes the forward in-kernel from z."""

    @staticmethod
    def forward(ctx, z, tree):
        handle = tree.device_handle(z.device.index)
        ctx.tree = tree
        ctx.save_for_backward(z)
        return _C.soft_forward(handle, z)

    @staticmethod
    def backward(ctx, gP):
        (z,) = ctx.saved_tensors
        handle = ctx.tree.device_handle(z.device.index)
        gz = _C.soft_backward(handle, z, gP)
        return gz.to(z.dtype), None


class _NodeLogitsFn(torch.autograd.Function):
    """[B,R] child logits of every inner node; the VJP is one scatter-free gather kernel."""

    @staticmethod
    def forward(ctx, z, tree):
        handle = tree.device_handle(z.device.index)
        ctx.tree, ctx.z_dtype, ctx.dev = tree, z.dtype, z.device.index
        return _C.node_logits(handle, z)

    @staticmethod
    def backward(ctx, gs):
        handle = ctx.tree.device_handle(ctx.dev)
        return _C.node_logits_backward(handle, gs).to(ctx.z_dtype), None


class EmbeddedDecisionRules(nn.Module):
    """reference nbdt/model.py:65-123."""

    def __init__(self, dataset=None, path_graph=None, path_wnids=None, classes=(), hierarchy=None,
                 tree=None):
        super().__init__()
        if not tree:
            tree = Tree(dataset, path_graph, path_wnids, classes, hierarchy=hierarchy)
        self.tree = tree
        self.correct = 0
        self.total = 0
        self.I = torch.eye(len(self.tree.classes))

    @staticmethod
    def get_node_logits(outputs, node=None, new_to_old_classes=None, num_classes=None):
        """Child logits of ONE node: mean of the class logits under each child (reference :83-99).

        Kept for API parity (analysis code calls it per node).  It runs the fused node kernel for
        the node's hierarchy and slices out this node's columns; callers that need every node
        should use ``forward_nodes``.
        """
        assert node or (new_to_old_classes and num_classes), \
            "Either pass node or (new_to_old_classes mapping and num_classes)"
        if node is None:
            raise NotImplementedError(
                "get_node_logits without a node (ad-hoc class mapping) is not part of the HIP path")
        tree = node.tree
        _C.require_gpu(outputs, "get_node_logits")
        logits = _NodeLogitsFn.apply(outputs, tree)   # differentiable, like the reference's mean
        n = tree.flat.inode_wnids.index(node.wnid)
        b, e = int(tree.flat.node_off[n]), int(tree.flat.node_off[n + 1])
        return logits[:, b:e]

    @classmethod
    def get_all_node_outputs(cls, outputs, nodes):
        """reference :101-120 -- dict wnid -> {logits, preds, probs, entropy}."""
        if not nodes:
            return {}
        tree = nodes[0].tree
        _C.require_gpu(outputs, "forward_nodes")
        handle = tree.device_handle(outputs.device.index)
        logits, probs, preds, ent = _C.node_outputs(handle, outputs)
        flat = tree.flat
        index = {w: i for i, w in enumerate(flat.inode_wnids)}
        out = {}
        for node in nodes:
            n = index[node.wnid]
            b, e = int(flat.node_off[n]), int(flat.node_off[n + 1])
            out[node.wnid] = {
                "logits": logits[:, b:e],
                "preds": preds[:, n],
                "probs": probs[:, b:e],
                "entropy": ent[:, n],
            }
        return out

    def forward_nodes(self, outputs):
        return self.get_all_node_outputs(outputs, self.tree.inodes)


class HardEmbeddedDecisionRules(EmbeddedDecisionRules):
    """Greedy root->leaf traversal (reference :126-203); output is one-hot, detached."""

    @classmethod
    def get_node_logits_filtered(cls, node, outputs, targets):
        """reference :127-143 -- rows whose label lies under `node`, their child logits and the
        child index of each label (`class_index_to_child_index[t][0]`)."""
        classes = [node.class_index_to_child_index[int(t)] for t in targets]
        selector = [bool(c) for c in classes]
        targets_sub = [c[0] for c in classes if c]
        outputs = outputs[torch.tensor(selector, dtype=torch.bool, device=outputs.device)]
        if outputs.size(0) == 0:
            return selector, outputs[:, : node.num_classes], targets_sub
        outputs_sub = cls.get_node_logits(outputs, node)
        return selector, outputs_sub, targets_sub

    @classmethod
    def traverse_tree(cls, wnid_to_outputs, tree):
        """reference :146-187 -- (predicted classes, decisions) from a dict of per-node outputs a caller built or edited
        itself (``forward_nodes`` / ``get_all_node_outputs``).  API parity only: ``forward`` / ``forward_with_decisions``
        never build the dict -- they are one fused kernel launch -- so this host walk is for analysis code that
        intervenes on node outputs.  A sample whose walk reaches an inner node missing from the dict raises KeyError
        (the reference dereferences ``None`` there)."""
        first = wnid_to_outputs[tree.inodes[0].wnid]["logits"]
        host = {w: (o["preds"].detach().cpu().tolist(), o["probs"].detach().cpu(), o["entropy"].detach().cpu())
                for w, o in wnid_to_outputs.items()}
        preds, decisions = [], []
        for i in range(int(first.shape[0])):
            node, walk = tree.root, [{"node": tree.root, "name": "root", "prob": 1, "entropy": 0}]
            while not node.is_leaf():
                choice, probs, entropy = host[node.wnid]
                k = int(choice[i])
                node = node.children[k]
                walk.append({"node": node, "name": node.name, "prob": float(probs[i][k]), "next_index": k,
                             "entropy": float(entropy[i])})
            preds.append(tree.wnid_to_class_index[node.wnid])
            decisions.append(walk)
        return torch.tensor(preds, dtype=torch.long, device=first.device), decisions

    def predicted_to_logits(self, predicted):
        if self.I.device != predicted.device:
            self.I = self.I.to(predicted.device)
        return self.I[predicted]

    def _decisions_to_python(self, pred, bufs):
        """Device decision buffers -> the reference's list-of-dicts format (:165-185)."""
        pn, pc, pp, pe = (b.cpu() for b in bufs)
        inodes = self.tree.inodes
        decisions = []
        for i in range(pn.shape[0]):
            decision = [{"node": self.tree.root, "name": "root", "prob": 1, "entropy": 0}]
            for d in range(pn.shape[1]):
                n = int(pn[i, d])
                if n < 0:
                    break
                k = int(pc[i, d])
                child = inodes[n].children[k]
                decision.append({"node": child, "name": child.name, "prob": float(pp[i, d]),
                                 "next_index": k, "entropy": float(pe[i, d])})
            decisions.append(decision)
        return decisions

    def forward_with_decisions(self, outputs):
        _C.require_gpu(outputs, "HardEmbeddedDecisionRules")
        handle = self.tree.device_handle(outputs.device.index)
        pred, onehot, bufs = _C.hard_forward(handle, outputs.detach(), want_onehot=True,
                                             want_decisions=True)
        decisions = self._decisions_to_python(pred, bufs)
        onehot._nbdt_output_flag = True  # checked in nbdt losses, to prevent mistakes
        return onehot, decisions

    def forward(self, outputs):
        _C.require_gpu(outputs, "HardEmbeddedDecisionRules")
        handle = self.tree.device_handle(outputs.device.index)
        _, onehot, _ = _C.hard_forward(handle, outputs.detach(), want_onehot=True)
        onehot._nbdt_output_flag = True
        return onehot

    def predict(self, outputs):
        """[B] int64 predicted classes without materialising the one-hot matrix."""
        handle = self.tree.device_handle(outputs.device.index)
        pred, _, _ = _C.hard_forward(handle, outputs.detach(), want_onehot=False)
        return pred


class SoftEmbeddedDecisionRules(EmbeddedDecisionRules):
    """Path-probability product over the hierarchy (reference :206-273); differentiable."""

    @classmethod
    def traverse_tree(cls, wnid_to_outputs, tree):
        """reference :208-242 -- class probabilities as the product, over the inner nodes, of the probability of the
        child each class lies under, from a caller-built dict of per-node outputs.  API parity only (``forward`` is the
        fused kernel); plain tensor ops on whatever device the dict's tensors live on."""
        first = wnid_to_outputs[tree.inodes[0].wnid]["logits"]
        class_probs = torch.ones((first.shape[0], len(tree.classes)), device=first.device)
        for node in tree.inodes:
            pairs = [(old, new) for new, olds in node.child_index_to_class_index.items() for old in olds]
            olds, news = [p[0] for p in pairs], [p[1] for p in pairs]
            assert len(set(olds)) == len(olds), "a class under two children of one node"
            class_probs[:, olds] *= wnid_to_outputs[node.wnid]["probs"][:, news].to(class_probs.dtype)
        return class_probs

    def forward_with_decisions(self, outputs):
        """reference :244-266.  The reference reports sample 0's node probabilities for every
        sample (`_out["probs"][0]`, :259 -- SURVEY 8c calls it a bug); here each sample gets its
        own probabilities/entropies."""
        wnid_to_outputs = self.forward_nodes(outputs)
        out = self.forward(outputs)
        _, predicted = out.max(1)
        predicted = predicted.cpu()
        host = {w: {"probs": o["probs"].cpu(), "entropy": o["entropy"].cpu()}
                for w, o in wnid_to_outputs.items()}
        leaf_to_steps = self.tree.get_leaf_to_steps()
        decisions = []
        for index, prediction in enumerate(predicted):
            leaf = self.tree.wnids_leaves[int(prediction)]
            steps = [dict(s) for s in leaf_to_steps[leaf]]
            probs, entropies = [1], [0]
            for step in steps[:-1]:
                o = host[step["node"].wnid]
                probs.append(float(o["probs"][index][step["next_index"]]))
                entropies.append(float(o["entropy"][index]))
            for step, prob, entropy in zip(steps, probs, entropies):
                step["prob"] = float(prob)
                step["entropy"] = float(entropy)
            decisions.append(steps)
        return out, decisions

    def forward(self, outputs, wnid_to_outputs=None):
        _C.require_gpu(outputs, "SoftEmbeddedDecisionRules")
        logits = _SoftRulesFn.apply(outputs, self.tree)
        logits._nbdt_output_flag = True  # checked in nbdt losses, to prevent mistakes
        return logits


def coerce_state_dict(state_dict, reference_state_dict):
    """reference nbdt/models/utils.py:65-76: unwrap {'net': ...} and fix the `module.` prefix."""
    if "net" in state_dict:
        state_dict = state_dict["net"]
    has_reference_module = list(reference_state_dict)[0].startswith("module.")
    has_module = list(state_dict)[0].startswith("module.")
    if not has_reference_module and has_module:
        state_dict = {k.replace("module.", "", 1): v for k, v in state_dict.items()}
    elif has_reference_module and not has_module:
        state_dict = {"module." + k: v for k, v in state_dict.items()}
    return state_dict


class NBDT(nn.Module):
    """Backbone + decision rules (reference nbdt/model.py:281-361)."""

    def __init__(self, dataset, model, arch=None, path_graph=None, path_wnids=None, classes=None,
                 hierarchy=None, pretrained=None, **kwargs):
        super().__init__()
        if dataset and not hierarchy and not path_graph:
            assert arch, "Must specify `arch` if no `hierarchy` or `path_graph`"
            hierarchy = f"induced-{arch}"
        if pretrained and not arch:
            raise UserWarning(
                "To load a pretrained NBDT, you need to specify the `arch`. "
                "`arch` is the name of the architecture. e.g., ResNet18")
        if isinstance(model, str):
            raise NotImplementedError("Model must be nn.Module")
        tree = Tree(dataset, path_graph, path_wnids, classes, hierarchy=hierarchy)
        self.init(dataset, model, tree, arch=arch, pretrained=pretrained, hierarchy=hierarchy, **kwargs)

    def init(self, dataset, model, tree, arch=None, pretrained=False, hierarchy=None, eval=True,
             Rules=HardEmbeddedDecisionRules):
        self.rules = Rules(tree=tree)
        self.model = model
        if pretrained:
            raise NotImplementedError(
                "pretrained=True downloads release checkpoints (reference :337-341); there is no "
                "network on the target -- load a local checkpoint with load_state_dict instead")
        if eval:
            self.eval()

    def load_state_dict(self, state_dict, **kwargs):
        state_dict = coerce_state_dict(state_dict, self.model.state_dict())
        return self.model.load_state_dict(state_dict, **kwargs)

    def state_dict(self, *args, **kwargs):
        return self.model.state_dict(*args, **kwargs)

    def forward(self, x):
        x = self.model(x)
        x = self.rules(x)
        return x

    def forward_with_decisions(self, x):
        x = self.model(x)
        x, decisions = self.rules.forward_with_decisions(x)
        return x, decisions


class HardNBDT(NBDT):
    def __init__(self, *args, **kwargs):
        kwargs.update({"Rules": HardEmbeddedDecisionRules})
        super().__init__(*args, **kwargs)


class SoftNBDT(NBDT):
    def __init__(self, *args, **kwargs):
        kwargs.update({"Rules": SoftEmbeddedDecisionRules})
        super().__init__(*args, **kwargs)
