"""Path helpers and dataset constants for the NBDT hot path.

Only the pieces of the reference's ``nbdt/utils.py`` the hot path touches are
mirrored (hierarchy name -> file, dataset -> class count; reference
nbdt/utils.py:20-40, 62-71).  Progress bars, colour printing, flag plumbing and
image download helpers are out of scope.
"""
import os
from pathlib import Path

METHODS = ("wordnet", "random", "induced")
DATASETS = ("CIFAR10", "CIFAR100", "TinyImagenet200", "Imagenet1000")
DATASET_TO_NUM_CLASSES = {
    "CIFAR10": 10,
    "CIFAR100": 100,
    "TinyImagenet200": 200,
    "Imagenet1000": 1000,
}


def fwd():
    """Directory of this package (where hierarchies/ and wnids/ live)."""
    return Path(__file__).parent.absolute()


def hierarchy_to_path_graph(dataset, hierarchy):
    return os.path.join(fwd(), f"hierarchies/{dataset}/graph-{hierarchy}.json")


def dataset_to_default_path_graph(dataset):
    return hierarchy_to_path_graph(dataset, "induced")


def dataset_to_default_path_wnids(dataset):
    return os.path.join(fwd(), f"wnids/{dataset}.txt")
