"""WideResNet-28-10 -- reference nbdt/models/wideresnet.py:1-5 (pytorchcv wrn28_10_cifar10/100) and
:28-40 (``wrn28_10``: same net with a global average pool, for 64x64 TinyImagenet inputs; the
engine's head is a global pool for any input size, so the three factories share one implementation)."""
from nbdt.engine import WRNEngine
from nbdt.models._hip_module import HipBackbone


def _wrn(num_classes, pretrained=False, progress=True, dataset="CIFAR10", device="cuda", seed=0, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained checkpoints need network access; use load_state_dict")
    return HipBackbone(WRNEngine(num_classes=num_classes, blocks=28, width_factor=10, device=device, seed=seed))


def wrn28_10_cifar10(num_classes=10, **kwargs):
    return _wrn(num_classes, **kwargs)


def wrn28_10_cifar100(num_classes=100, **kwargs):
    return _wrn(num_classes, **kwargs)


def wrn28_10(num_classes=10, **kwargs):
    return _wrn(num_classes, **kwargs)
