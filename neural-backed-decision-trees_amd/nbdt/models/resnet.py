"""CIFAR ResNets (BasicBlock variants) -- reference nbdt/models/resnet.py:42-74, 115-149, 161-199.
Bottleneck variants (ResNet50+) are not on the hot path configs and are not built."""
from nbdt.engine import ResNetEngine
from nbdt.models._hip_module import HipBackbone


def _resnet(num_blocks, num_classes=10, pretrained=False, progress=True, dataset="CIFAR10", device="cuda", seed=0):
    if pretrained:
        raise NotImplementedError("pretrained checkpoints need network access; use load_state_dict")
    return HipBackbone(ResNetEngine(num_classes=num_classes, num_blocks=num_blocks, device=device, seed=seed))


def ResNet10(**kwargs):
    return _resnet((1, 1, 1, 1), **kwargs)


def ResNet18(**kwargs):
    return _resnet((2, 2, 2, 2), **kwargs)


def ResNet34(**kwargs):
    return _resnet((3, 4, 6, 3), **kwargs)
