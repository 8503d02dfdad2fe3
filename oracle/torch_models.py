"""CPU ORACLE backbones in plain PyTorch fp32 -- TEST INFRASTRUCTURE ONLY (see nbdt_oracle.py).

* ``WRN`` restates pytorchcv's ``CIFARWRN`` (``get_wrn_cifar(blocks=28, width_factor=10)``), the
  third-party model behind the reference's ``nbdt/models/wideresnet.py:1-5, 28-40``.  pytorchcv is
  un-vendored and unpinned (reference requirements.txt:1) and absent from this box, so the
  restatement follows its published architecture: ``features.init_block`` conv3x3(3->16); three
  stages of ``PreResUnit``s (BN-ReLU-conv3x3 twice; 1x1 ``identity_conv`` applied to the
  PRE-ACTIVATED input when the shape changes; stride 2 at the first unit of stages 2 and 3);
  ``features.post_activ`` BN+ReLU; 8x8 average pool; ``output`` Linear.  Convs are bias-free and
  initialised with ``kaiming_uniform_``.  State-dict keys match pytorchcv's.
  PARITY UNPINNED for this backbone: no reference test or golden vector pins pytorchcv's numerics;
  what is pinned is the parameter count (36.45 M conv parameters for WRN-28-10) and the key names.
* ``ResNet18`` restates the reference's own CIFAR ResNet (nbdt/models/resnet.py:42-74, 115-149).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _RoundBF16(torch.autograd.Function):
    """Identity that rounds to bf16 in BOTH directions: emulates storing an activation (forward) and
    its gradient (backward) in bf16 HBM buffers, as the HIP engine does.  fp32 arithmetic otherwise."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundFwdBF16(torch.autograd.Function):
    """bf16 rounding of a weight as the conv kernels see it; gradient passes through in fp32."""

    @staticmethod
    def forward(ctx, w):
        return w.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


_EMULATE = [False]


class emulate_bf16:
    """Context manager: inside it the oracle backbones round conv weights, conv outputs and
    BN/ReLU outputs (and the matching gradients) to bf16 -- the storage points of the HIP engine."""

    def __enter__(self):
        self.prev = _EMULATE[0]
        _EMULATE[0] = True

    def __exit__(self, *a):
        _EMULATE[0] = self.prev


def _q(x):
    return _RoundBF16.apply(x) if _EMULATE[0] else x


def _conv(conv, x):
    if not _EMULATE[0]:
        return conv(x)
    return F.conv2d(x, _RoundFwdBF16.apply(conv.weight), None, conv.stride, conv.padding)


class _PreConv(nn.Module):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.bn = nn.BatchNorm2d(cin)
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)

    def forward(self, x):
        a = _q(F.relu(self.bn(x)))
        return _q(_conv(self.conv, a)), a


class _PreResBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = _PreConv(cin, cout, 3, stride)
        self.conv2 = _PreConv(cout, cout, 3, 1)

    def forward(self, x):
        x, pre = self.conv1(x)
        x, _ = self.conv2(x)
        return x, pre


class _PreResUnit(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.resize = cin != cout or stride != 1
        self.body = _PreResBlock(cin, cout, stride)
        if self.resize:
            self.identity_conv = nn.Conv2d(cin, cout, 1, stride=stride, bias=False)

    def forward(self, x):
        identity = x
        x, pre = self.body(x)
        if self.resize:
            identity = _q(_conv(self.identity_conv, pre))
        return _q(x + identity)


class _PostActiv(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.bn = nn.BatchNorm2d(c)

    def forward(self, x):
        return F.relu(self.bn(x))


class WRN(nn.Module):
    def __init__(self, num_classes=10, blocks=28, width_factor=10):
        super().__init__()
        n = (blocks - 4) // 6
        widths = [16 * width_factor, 32 * width_factor, 64 * width_factor]
        feats = nn.Sequential()
        feats.add_module("init_block", nn.Conv2d(3, 16, 3, padding=1, bias=False))
        cin = 16
        for i, cout in enumerate(widths):
            stage = nn.Sequential()
            for j in range(n):
                stage.add_module(f"unit{j + 1}", _PreResUnit(cin, cout, 2 if (j == 0 and i != 0) else 1))
                cin = cout
            feats.add_module(f"stage{i + 1}", stage)
        feats.add_module("post_activ", _PostActiv(cin))
        feats.add_module("final_pool", nn.AdaptiveAvgPool2d(1))
        self.features = feats
        self.output = nn.Linear(cin, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight)

    def forward(self, x):
        if _EMULATE[0]:   # the stem output is the first bf16 storage point
            f = self.features
            h = _q(f.init_block(x))      # the stem kernel reads fp32 weights; only its output is bf16
            for name, m in f.named_children():
                if name != "init_block":
                    h = m(h)
            return self.output(h.flatten(1))
        return self.output(self.features(x).flatten(1))


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.shortcut = nn.Sequential()
        if stride != 1 or cin != cout:
            self.shortcut = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                                          nn.BatchNorm2d(cout))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + self.shortcut(x))


class ResNet18(nn.Module):
    def __init__(self, num_classes=10):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (cout, stride) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)]):
            blocks = [_BasicBlock(cin, cout, stride), _BasicBlock(cout, cout, 1)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
            cin = cout
        self.linear = nn.Linear(512, num_classes)

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        for i in range(4):
            out = getattr(self, f"layer{i + 1}")(out)
        out = F.avg_pool2d(out, out.size()[2:]).flatten(1)
        return self.linear(out)


# --------------------------------------------------------------------------------------------------
# EfficientNet-B0: restates pytorchcv's ``efficientnet_b0`` (``get_efficientnet(
# version="b0", in_size=(224, 224))``, non-TF mode: symmetric padding, bn_eps 1e-5), the third-party
# model the reference re-exports at nbdt/models/__init__.py:3.  PARITY UNPINNED (pytorchcv absent,
# unpinned, no golden vectors); pinned here: the published architecture, 5,288,548 parameters for
# 1000 classes, and pytorchcv's state-dict key names (``output.fc.weight`` is the classifier key
# the reference itself relies on, nbdt/graph.py:393).

class _Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class _ConvBlock(nn.Module):
    """pytorchcv ConvBlock: conv (bias-free) + BatchNorm + optional Swish.
    Under emulate_bf16 the raw conv output is rounded (every engine stores it), the 1x1 convs read bf16 weights
    (`w_bf16`; the stem and the depthwise kernels read the fp32 masters), and the block's output is rounded only where
    the EfficientNet engine stores it (`round_out`: not after a depthwise conv -- swish(bn(.)) exists only inside its
    kernels -- nor after the project conv, whose output is stored after the skip add, nor after the final block)."""

    def __init__(self, cin, cout, k, stride=1, groups=1, act=True, round_out=True, w_bf16=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=1e-5)
        self.act = act
        self.round_out = round_out
        self.w_bf16 = w_bf16 and groups == 1

    def forward(self, x):
        x = _q(_conv(self.conv, x) if self.w_bf16 else self.conv(x))
        x = self.bn(x)
        x = x * torch.sigmoid(x) if self.act else x
        return _q(x) if self.round_out else x


class _SEBlock(nn.Module):
    def __init__(self, channels, mid):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, mid, 1, bias=True)
        self.conv2 = nn.Conv2d(mid, channels, 1, bias=True)

    def forward(self, x):
        w = x.mean((2, 3), keepdim=True)
        w = self.conv1(w)
        w = w * torch.sigmoid(w)
        w = torch.sigmoid(self.conv2(w))
        return _q(x * w)                 # (the gated activation is what the engine stores: d_se)


class _EffiDwsConvUnit(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.residual = cin == cout and stride == 1
        self.dw_conv = _ConvBlock(cin, cin, 3, stride, groups=cin, round_out=False)
        self.se = _SEBlock(cin, cin // 4)
        self.pw_conv = _ConvBlock(cin, cout, 1, act=False, round_out=False)

    def forward(self, x):
        y = self.pw_conv(self.se(self.dw_conv(x)))
        return _q(y + x if self.residual else y)


class _EffiInvResUnit(nn.Module):
    def __init__(self, cin, cout, k, stride, exp_factor, se_factor=4):
        super().__init__()
        self.residual = cin == cout and stride == 1
        mid = cin * exp_factor
        self.conv1 = _ConvBlock(cin, mid, 1)
        self.conv2 = _ConvBlock(mid, mid, k, stride, groups=mid, round_out=False)
        self.se = _SEBlock(mid, mid // (exp_factor * se_factor))
        self.conv3 = _ConvBlock(mid, cout, 1, act=False, round_out=False)

    def forward(self, x):
        y = self.conv3(self.se(self.conv2(self.conv1(x))))
        return _q(y + x if self.residual else y)


class _InitBlock(nn.Module):
    def __init__(self, cout):
        super().__init__()
        self.conv = _ConvBlock(3, cout, 3, stride=2, w_bf16=False)

    def forward(self, x):
        return self.conv(x)


EFFNET_B0_STAGES = [
    # per stage: list of (out_channels, kernel, expansion); stage stride applies to its first unit
    (1, [(16, 3, 1)]),
    (2, [(24, 3, 6), (24, 3, 6)]),
    (2, [(40, 5, 6), (40, 5, 6)]),
    (2, [(80, 3, 6)] * 3 + [(112, 5, 6)] * 3),
    (2, [(192, 5, 6)] * 4 + [(320, 3, 6)]),
]


class EfficientNetB0(nn.Module):
    def __init__(self, num_classes=1000, dropout_rate=0.2):
        super().__init__()
        feats = nn.Sequential()
        feats.add_module("init_block", _InitBlock(32))
        cin = 32
        for i, (stride, units) in enumerate(EFFNET_B0_STAGES):
            stage = nn.Sequential()
            for j, (cout, k, exp) in enumerate(units):
                s = stride if j == 0 else 1
                unit = _EffiDwsConvUnit(cin, cout, s) if i == 0 else _EffiInvResUnit(cin, cout, k, s, exp)
                stage.add_module(f"unit{j + 1}", unit)
                cin = cout
            feats.add_module(f"stage{i + 1}", stage)
        feats.add_module("final_block", _ConvBlock(cin, 1280, 1, round_out=False))
        feats.add_module("final_pool", nn.AdaptiveAvgPool2d(1))
        self.features = feats
        self.output = nn.Sequential()
        if dropout_rate > 0.0:
            self.output.add_module("dropout", nn.Dropout(p=dropout_rate))
        self.output.add_module("fc", nn.Linear(1280, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return self.output(self.features(x).flatten(1))
