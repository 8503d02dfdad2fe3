"""``generate_hierarchy`` -- reference nbdt/hierarchy.py:59-127 for the ``induced`` method (the one every
shipped NBDT hierarchy uses).  ``wordnet`` needs the WordNet corpus (nltk, absent on the target image); ``random``
ablation hierarchies and ``--extra`` augmentation are out of scope: GENERATING them
raises here.  Hierarchy files of those kinds written by the reference still load (nbdt/tree.py), which is why
nbdt.graph.generate_graph_fname keeps their naming scheme."""
from nbdt.graph import build_induced_graph, get_graph_path_from_args, write_graph
from nbdt.tree import get_wnids
from nbdt.utils import dataset_to_default_path_wnids


def generate_hierarchy(dataset, method, seed=0, branching_factor=2, extra=0, no_prune=False, fname="", path="",
                       single_path=False, induced_linkage="ward", induced_affinity="euclidean", checkpoint=None,
                       arch=None, model=None, path_wnids=None, **kwargs):
    if method != "induced":
        raise NotImplementedError(f'Method "{method}" is not built (only induced hierarchies are in scope)')
    if extra:
        raise NotImplementedError("graph augmentation (--extra) is not built")
    wnids = get_wnids(path_wnids or dataset_to_default_path_wnids(dataset))
    G = build_induced_graph(wnids, dataset=dataset, checkpoint=checkpoint,
                            model=None if model is not None else arch, linkage=induced_linkage,
                            affinity=induced_affinity, branching_factor=branching_factor,
                            state_dict=model.state_dict() if model is not None else None)
    assert all(w in G.nodes for w in wnids)
    # The reference prunes single-successor nodes here unless --no-prune (nbdt/hierarchy.py:96-98).  Agglomerative
    # clustering merges exactly two clusters per step whatever the linkage, so an induced hierarchy never has such a
    # node and pruning is the identity (no_prune only shows up in the file name) -- asserted rather than assumed:
    inner = [n for n in G.nodes if n not in set(wnids)]
    assert all(len(list(G.succ(n))) == 2 for n in inner), "induced hierarchy with a non-binary inner node"
    path = get_graph_path_from_args(dataset=dataset, method=method, seed=seed, branching_factor=branching_factor,
                                    extra=extra, no_prune=no_prune, fname=fname, path=path,
                                    multi_path=single_path, induced_linkage=induced_linkage,
                                    induced_affinity=induced_affinity, checkpoint=checkpoint, arch=arch)
    write_graph(G, path)
    print("==> Wrote tree to {}".format(path))
    return path
