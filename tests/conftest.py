import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbdt_path  # noqa: E402

nbdt_path.add(oracle=True)

GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG = os.path.join(nbdt_path.PKG_DIR, "nbdt")

# tag -> (dataset, hierarchy): the cases tests/golden/make_golden.py recorded from the reference
GOLDEN_CASES = {
    "cifar10_wrn": ("CIFAR10", "induced-wrn28_10_cifar10"),
    "cifar10_r18": ("CIFAR10", "induced-ResNet18"),
    "cifar10_wordnet": ("CIFAR10", "wordnet"),
    "cifar100_wrn": ("CIFAR100", "induced-wrn28_10_cifar100"),
    "cifar100_wordnet": ("CIFAR100", "wordnet"),
    "tiny_r18": ("TinyImagenet200", "induced-ResNet18"),
    "tiny_wordnet": ("TinyImagenet200", "wordnet"),
    "imagenet_eff": ("Imagenet1000", "induced-efficientnet_b7b"),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def pkg_dir():
    return PKG
