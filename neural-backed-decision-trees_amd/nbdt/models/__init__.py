"""Backbones of the NBDT hot path, MI355X-native (same factory names as the reference's
``nbdt.models``: resnet.py :171-179, wideresnet.py :1-5, 28-40, __init__.py :3 efficientnet_b0).  Each factory returns an
``nn.Module`` facade (see _hip_module.py) over the HIP execution engine."""
from .efficientnet import efficientnet_b0
from .resnet import ResNet10, ResNet18, ResNet34
from .wideresnet import wrn28_10, wrn28_10_cifar10, wrn28_10_cifar100

__all__ = ("ResNet10", "ResNet18", "ResNet34", "wrn28_10", "wrn28_10_cifar10", "wrn28_10_cifar100",
           "efficientnet_b0")


def get_model_choices():
    return list(__all__)
