"""ResNet-18/34/50/101/152 in plain torch.nn with torchvision-identical topology, parameter
names and state_dict order (torchvision is not installed on the target image).

The reference builds its encoder from `torchvision.models.resnet*` (resnet_model.py:15,31-43).
Per BASELINE.json:north_star the convolutional backbone stays on PyTorch-ROCm (MIOpen); this file
only restates the architecture: 7x7/2 conv + BN + ReLU + 3x3/2 max-pool, BasicBlock [2,2,2,2] /
[3,4,6,3], Bottleneck v1.5 (stride on the 3x3) [3,4,6,3] / [3,4,23,3] / [3,8,36,3], kaiming-normal
(fan_out) conv init, BN weight 1 / bias 0.  torchvision 0.8.0 itself is not importable here; the arithmetic is
pinned against an independent implementation of the same published architecture, `transformers.ResNetModel`
with these weights copied in block by block (1e-10 in float64, eval and train mode, running statistics
included: tests/test_host_logic.py::test_resnet_arithmetic_matches_an_independent_implementation), and the
layout by state_dict key/shape/parameter-count lists (test_resnet_state_dict_layout).
"""
from __future__ import annotations

from typing import List, Type, Union

import torch
from torch import Tensor, nn

from .bn2d import Conv2d, FusedBatchNormAct2d, checkpoint_block, fork_conv1x1


def conv3x3(cin, cout, stride=1):
    return Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


def conv1x1(cin, cout, stride=1):
    return Conv2d(cin, cout, 1, stride=stride, bias=False)


def _conv(conv, x, bn, sole_consumer=False):
    """conv(x), telling an in-tree Conv2d which BatchNorm consumes the result and whether it is the ONLY reader of x
    (then its input-gradient GEMM may also perform the backward reduction of the BatchNorm that produced x)."""
    return conv(x, stats_for=bn, sole_consumer=sole_consumer) if isinstance(conv, Conv2d) else conv(x)


def _bn(bn, x, residual=None, relu=False, consumer=None):
    """bn -> (+residual) -> (relu): one fused HIP pass when `bn` is a FusedBatchNormAct2d, the stock
    three ops otherwise (any other norm_layer).  consumer: the layer that is the only reader of the result (a block's last
    BatchNorm pass computes the shortcut's BatchNorm itself: FusedBatchNormAct2d.forward)."""
    if isinstance(bn, FusedBatchNormAct2d):
        return bn(x, residual, relu, consumer=consumer if isinstance(consumer, (Conv2d, FusedBatchNormAct2d)) else None)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    checkpoint = False  # keep only the block input for backward and re-run the block there (set per instance)

    def forward(self, x: Tensor) -> Tensor:
        return checkpoint_block(self._run, x) if self.checkpoint else self._run(x)

    def _run(self, x: Tensor) -> Tensor:
        ds = self.downsample
        if isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[1], FusedBatchNormAct2d):
            # (ds[0] / ds[1] are called directly: forward hooks on the `downsample` Sequential ITSELF do not fire on this path --
            # hooks on its convolution and BatchNorm do; state_dict keys are unchanged)
            identity = _bn(ds[1], ds[0](x), consumer=self.bn2)                 # (its apply is left to bn2's pass, the only reader)
        else:
            identity = x if ds is None else ds(x)
        # x feeds conv1 AND the shortcut: conv1's input gradient is only one of the two terms of x's gradient
        out = _bn(self.bn1, _conv(self.conv1, x, self.bn1, sole_consumer=False), relu=True)
        return _bn(self.bn2, _conv(self.conv2, out, self.bn2, sole_consumer=True), identity, relu=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = conv1x1(inplanes, planes)
        self.conv1.fork_entry = True  # x feeds conv1 AND the residual branch (identity or downsample)
        self.bn1 = norm_layer(planes)
        self.conv2 = conv3x3(planes, planes, stride)  # v1.5: stride on the 3x3
        self.bn2 = norm_layer(planes)
        self.conv3 = conv1x1(planes, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    checkpoint = False

    def forward(self, x: Tensor) -> Tensor:
        return checkpoint_block(self._run, x) if self.checkpoint else self._run(x)

    def _run(self, x: Tensor) -> Tensor:
        # x has two consumers; in backward "conv1's input gradient + the residual branch's gradient" is one GEMM
        # (the 1x1 convolutions that run as in-tree GEMMs sum the statistics of the BatchNorm behind them in their epilogue)
        out, identity = fork_conv1x1(self.conv1, x, stats_for=self.bn1)
        if self.downsample is not None:
            ds = self.downsample
            if isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[1], FusedBatchNormAct2d):
                # (conv -> bn: statistics from the GEMM where in-tree; the apply is left to bn3's pass, the only reader)
                identity = _bn(ds[1], _conv(ds[0], identity, ds[1]), consumer=self.bn3)
            else:
                identity = ds(identity)
        out = _bn(self.bn1, out, relu=True)
        out = _bn(self.bn2, _conv(self.conv2, out, self.bn2, sole_consumer=True), relu=True, consumer=self.conv3)
        return _bn(self.bn3, _conv(self.conv3, out, self.bn3, sole_consumer=True), identity, relu=True)


class ResNet(nn.Module):
    def __init__(self, block: Type[Union[BasicBlock, Bottleneck]], layers: List[int], num_classes: int = 1000,
                 norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or FusedBatchNormAct2d  # an nn.BatchNorm2d (same state_dict) that can fuse
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.conv1 = Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       self._norm_layer(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, self._norm_layer)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, norm_layer=self._norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x: Tensor) -> Tensor:
        t = _bn(self.bn1, _conv(self.conv1, x, self.bn1), relu=True)
        x = self.maxpool(t)
        tag = getattr(t, "_peclr_absmax", None)
        if tag is not None and tag[1] == t._version and isinstance(self.maxpool, nn.MaxPool2d):
            # (the rectified tensor is non-negative and every element lies in some pooling window: the maximum carries over exactly)
            x._peclr_absmax = (tag[0], x._version)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


_SPECS = {"resnet18": (BasicBlock, [2, 2, 2, 2]), "resnet34": (BasicBlock, [3, 4, 6, 3]),
          "resnet50": (Bottleneck, [3, 4, 6, 3]), "resnet101": (Bottleneck, [3, 4, 23, 3]),
          "resnet152": (Bottleneck, [3, 8, 36, 3])}


def _make(name: str, pretrained: Union[bool, str] = False, norm_layer=None, **kw) -> ResNet:
    block, layers = _SPECS[name]
    model = ResNet(block, layers, norm_layer=norm_layer, **kw)
    if isinstance(pretrained, str):  # explicit weight file (no network on the target)
        model.load_state_dict(torch.load(pretrained, map_location="cpu"))
    return model


def set_activation_checkpointing(module: nn.Module, enabled: bool = True) -> int:
    """Every residual block under `module` keeps only its input for the backward pass and recomputes its
    convolutions / BatchNorms there (one extra forward, ~1/3 more compute) -- activation memory drops from
    "every tensor between two convolutions" to "one tensor per block".  Returns the number of blocks switched."""
    n = 0
    for m in module.modules():
        if isinstance(m, (BasicBlock, Bottleneck)):
            m.checkpoint = enabled
            n += 1
    return n


def resnet18(pretrained=False, **kw): return _make("resnet18", pretrained, **kw)
def resnet34(pretrained=False, **kw): return _make("resnet34", pretrained, **kw)
def resnet50(pretrained=False, **kw): return _make("resnet50", pretrained, **kw)
def resnet101(pretrained=False, **kw): return _make("resnet101", pretrained, **kw)
def resnet152(pretrained=False, **kw): return _make("resnet152", pretrained, **kw)


def conv_flops_per_image(model: nn.Module, hw=(224, 224)) -> int:
    """2*MACs of every Conv2d/Linear for one forward pass of one image (hook-based counter);
    bench.py uses it for the end-to-end MFMA roofline of the backbone."""
    total = [0]

    def conv_hook(m, inp, out):
        kh, kw = m.kernel_size
        total[0] += 2 * out.numel() // out.shape[0] * (m.in_channels // m.groups) * kh * kw

    def lin_hook(m, inp, out):
        total[0] += 2 * m.in_features * m.out_features

    hooks = []
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, nn.Linear):
            hooks.append(m.register_forward_hook(lin_hook))
    was = model.training
    model.eval()
    with torch.no_grad():
        dev = next(model.parameters()).device
        model(torch.zeros(1, 3, *hw, device=dev))
    model.train(was)
    for h in hooks:
        h.remove()
    return total[0]
