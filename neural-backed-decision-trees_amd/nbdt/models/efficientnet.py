"""EfficientNet-B0 -- the pytorchcv ``efficientnet_b0`` factory the reference re-exports at
nbdt/models/__init__.py:3 (README.md:141-151 uses it for ImageNet NBDTs)."""
from nbdt.engine_effnet import EfficientNetEngine
from nbdt.models._hip_module import HipBackbone


def efficientnet_b0(num_classes=1000, pretrained=False, in_size=(224, 224), dropout_rate=0.2, device="cuda",
                    seed=0, **kwargs):
    if pretrained:
        raise NotImplementedError("pretrained checkpoints need network access; use load_state_dict")
    return HipBackbone(EfficientNetEngine(num_classes=num_classes, dropout_rate=dropout_rate, device=device,
                                          seed=seed))
