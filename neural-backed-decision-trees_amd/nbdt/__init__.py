"""nbdt -- MI355X-native hot path of Neural-Backed Decision Trees.

Same import surface as the reference package for the hot path: ``nbdt.model`` (SoftNBDT,
HardNBDT, decision rules), ``nbdt.loss`` (SoftTreeSupLoss), ``nbdt.tree`` (Tree, Node),
``nbdt.models`` (ResNet18, wrn28_10_cifar10, ...).  All compute runs in libnbdt_hip.so
(hand-written HIP for gfx950); there is no CPU fallback.
"""
__version__ = "0.1.0"
