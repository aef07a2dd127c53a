"""Encoder wrapper: the reference's `ResNetModel` (resnet_model.py:6-55) and its factory
`get_wrapper_model` (utils.py:412-428), on the in-tree ResNet.

state_dict contract kept (needed by peclr_to_torchvision and the published checkpoints):
`features.{0,1,4,5,6,7}.*` = conv1, bn1, layer1..4 in torchvision's order, plus the unused
`final_layer.0.{weight,bias}` = Linear(in_features, 21*3+1) that lives in the optimiser but
never receives a gradient in "pretraining" mode (resnet_model.py:27-29,51-52).
"""
from __future__ import annotations

import os
import warnings

from torch import nn

from . import resnet as _resnet
from .bn2d import Conv2d, FusedBatchNormAct2d
from .config import Config


class TailAvgPool(nn.AdaptiveAvgPool2d):
    """AdaptiveAvgPool2d((1, 1)) that lets an already pooled [N, C] tensor through unchanged."""

    def __init__(self):
        super().__init__(output_size=(1, 1))

    def forward(self, x):
        return x if x.dim() == 2 else super().forward(x)


class _Features(nn.Sequential):
    """nn.Sequential (same indices, same state_dict keys) that tells the stem convolution which BatchNorm consumes its output,
    so that an in-tree stem sums that layer's statistics in its epilogue (bn2d.Conv2d.forward(stats_for=...))."""

    def forward(self, x):
        mods = list(self)
        if len(mods) >= 2 and isinstance(mods[0], Conv2d) and isinstance(mods[1], FusedBatchNormAct2d):
            x = mods[1](mods[0](x, stats_for=mods[1]))
            mods = mods[2:]
        for m in mods:
            x = m(x)
        return x


class ResNetModel(nn.Module):
    def __init__(self, config, mode: str = ""):
        super().__init__()
        self.mode = mode
        resnet_name = config.model.backend_model.lower()
        model_function = self.get_resnet(resnet_name)
        # FusedBatchNormAct2d IS an nn.BatchNorm2d (the reference passes norm_layer=nn.BatchNorm2d):
        # identical parameters/buffers/state_dict; it can additionally run as one fused HIP pass
        model = model_function(pretrained=config.model.pretrained, norm_layer=FusedBatchNormAct2d)
        # features.2 (the stem ReLU) and features.3 (MaxPool2d(3, 2, 1)) are folded into features.1;
        # Identity modules keep the Sequential indices (and therefore every state_dict key) where the
        # reference has them.  Stock mode: F.batch_norm / relu / max_pool2d; HIP mode: one pass that never
        # writes the 112x112 activation.
        model.bn1.default_relu = True
        model.bn1.default_pool = True
        # features.8: the global average pool.  In HIP mode the last block's final BatchNorm performs it inside
        # its fused pass (bn + identity + ReLU + mean over H x W, fp32 [N, C] out: SURVEY.md section 8 f4) and
        # this module receives an already pooled 2-D tensor, which it passes through.
        last = model.layer4[-1]
        (last.bn3 if hasattr(last, "bn3") else last.bn2).tail_avgpool = True
        self.features = _Features(model.conv1, model.bn1, nn.Identity(), nn.Identity(), model.layer1,
                                      model.layer2, model.layer3, model.layer4, TailAvgPool())
        self.final_layer = nn.Sequential(nn.Linear(model.fc.in_features, 21 * 3 + 1))

    def get_resnet(self, resnet_name):
        table = {"resnet18": _resnet.resnet18, "resnet34": _resnet.resnet34, "resnet50": _resnet.resnet50,
                 "resnet101": _resnet.resnet101, "resnet152": _resnet.resnet152}
        if resnet_name not in table:
            raise NotImplementedError  # resnet_model.py:42-43
        return table[resnet_name]

    # where the trainer may cut the autograd graph for a staged backward: before layer4 (features[7]) and before
    # layer3 (features[6]) -- stage 0 = head + layer4, stage 1 = layer3, stage 2 = layer2 + layer1 + stem
    SEAMS = (7, 6)

    def forward(self, x, cut=None):
        """cut (optional): callable `cut(index, tensor) -> tensor` applied to the input of features[index] for every
        index in SEAMS -- the trainer uses it to split the autograd graph there (backward in stages, so that the
        gradient all-reduce of the later layers overlaps the backward of the earlier ones)."""
        if cut is None:
            z = self.features(x)
        else:
            z, lo = x, 0
            for at in sorted(self.SEAMS):
                z = cut(at, self.features[lo:at](z))
                lo = at
            z = self.features[lo:](z)
        z = z.flatten(start_dim=1)
        if self.mode == "pretraining":
            return z
        z = self.final_layer(z)
        return z[:, : 21 * 3], None, z[:, -1]


def get_wrapper_model(config, pretrained, wrapper: bool = False):
    """utils.py:412-428.  `pretrained=True` means ImageNet weights in the reference (a download);
    there is no network on the target, so it resolves to `config.pretrained_path` or
    $PECLR_IMAGENET_WEIGHTS/resnet<size>.pth when present and to random init (with a warning)
    otherwise."""
    if wrapper:
        raise NameError("WrapperModel is undefined in the reference as well (utils.py:425-426)")
    weights = False
    if pretrained:
        cand = config.get("pretrained_path") if hasattr(config, "get") else None
        if not cand and os.environ.get("PECLR_IMAGENET_WEIGHTS"):
            cand = os.path.join(os.environ["PECLR_IMAGENET_WEIGHTS"], f"resnet{config.resnet_size}.pth")
        if cand and os.path.exists(cand):
            weights = cand
        else:
            warnings.warn("pretrained ImageNet weights requested but no local weight file is available "
                          "(no network): the encoder is randomly initialised", stacklevel=2)
    cfg = Config({"model": {"backend_model": "resnet" + str(config.resnet_size), "norm_layer": "bn",
                            "use_var": False, "pretrained": weights},
                  "dataset": {"np": 21}, "loss": {"hmap": {"enabled": False}}})
    return ResNetModel(config=cfg, mode="pretraining")
