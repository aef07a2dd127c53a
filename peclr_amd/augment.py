"""Two-view augmentation producing the hybrid2 batch dict on the device.

Mirrors `Data_Set.prepare_hybrid2_sample` (reference src/data_loader/data_set.py:357-384) +
`SampleAugmenter.transform_sample` (src/data_loader/sample_augmenter.py:47-129) + ToTensor/Normalize
(src/data_loader/utils.py:283-293) for the published recipe (README.md: --color_jitter
--random_crop --rotate --crop -resize), split where the hardware wants it split:

  host   the PARAMETER side, per sample and in the reference's draw order from Python's `random`:
         angle, [crop margin], [jitter x, jitter y], [h, s, a, b]; crop box, rotation centre and
         matrix, jitter_x / jitter_y.  A few dozen scalar operations per sample on the very torch ops
         the reference uses (float32 tensor mean / max / pow), so that the drawn parameters equal the
         reference's for the same seed (pinned by tests/golden/g9_augment_params.json).
  device the PIXEL side for the whole batch and both views: csrc/augment.hip (two launches).

The emitted dict is what torch's default collate makes of the reference's per-sample dicts:
`transformed_image{1,2}` float32 [B,3,S,S]; `jitter_{x,y}_{1,2}` int64 [B]; `angle_{1,2}` float64 [B]
(only with rotate); `h/s/a/b_{1,2}` float64 [B] (only with colour jitter); `crop_margin_scale_{1,2}`
float64 [B]; `blur_flag_{1,2}` bool [B].

Flags outside the recipe (sobel_filter, cut_out, gaussian_blur, gaussian_noise, color_drop) raise
NotImplementedError; `resize` is required (without it the reference's crops have per-sample sizes
and cannot be collated either).
"""
from __future__ import annotations

import math
import random as _random
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _capi

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
PARENT_JOINT, CHILD_JOINT = 0, 2  # wrist, index_mcp (reference data_loader/utils.py:15-16)
_UNSUPPORTED = ("sobel_filter", "cut_out", "gaussian_blur", "gaussian_noise", "color_drop")

DEFAULT_PARAMS = {  # reference src/experiments/config/training_config.json
    "crop_margin": 1.25, "crop_margin_range": [0.9, 1.5], "hue_factor_range": [0.01, 1.0], "max_angle": 45,
    "min_angle": -45, "resize_shape": [128, 128], "sat_factor_range": [0.01, 1.0],
    "value_factor_alpha_range": [0.5, 1], "value_factor_beta_range": [5, 20], "crop_box_jitter": [0.0, 15.0],
}
RECIPE_FLAGS = {"color_jitter": True, "random_crop": True, "rotate": True, "crop": True, "resize": True}


def convert_to_2_5d(k: Tensor, joints3d: Tensor) -> Tuple[Tensor, Tensor]:
    """Pinhole projection + root-relative scaled depth (reference data_loader/utils.py:19-33)."""
    scale = (((joints3d[CHILD_JOINT] - joints3d[PARENT_JOINT]) ** 2).sum()) ** 0.5
    joints25d = ((k @ joints3d.T).T) / joints3d[:, -1:]
    joints25d[:, -1] = (joints3d[:, -1] - joints3d[PARENT_JOINT, -1]) / scale
    return joints25d, scale


def _rotation_matrix(center: Tuple[int, int], angle: float) -> List[List[float]]:
    """OpenCV's getRotationMatrix2D(center, angle, 1.0) (documented formula), float64."""
    a = angle * math.pi / 180.0
    al, be = math.cos(a), math.sin(a)
    return [[al, be, (1 - al) * center[0] - be * center[1]], [-be, al, be * center[0] + (1 - al) * center[1]]]


def _invert_affine(m: List[List[float]]) -> List[float]:
    """Destination -> source map, the way warpAffine derives it from the forward matrix."""
    (m0, m1, m2), (m3, m4, m5) = m
    d = m0 * m4 - m1 * m3
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m4 * d, m0 * d
    m0, m1, m3, m4 = a11, m1 * -d, m3 * -d, a22
    b1 = -m0 * m2 - m1 * m5
    b2 = -m3 * m2 - m4 * m5
    return [m0, m1, b1, m3, m4, b2]


class TwoViewAugmenter:
    def __init__(self, flags: Optional[Dict[str, bool]] = None, params: Optional[Dict] = None, rng=None,
                 channels_last: bool = True):
        """flags / params: the reference's `augmentation_flags` / `augmentation_params` (missing flags
        are off, missing params take training_config.json's values).  rng: object with `.uniform`
        (default: Python's global `random`, the generator the reference draws from)."""
        self.flags = dict(RECIPE_FLAGS if flags is None else flags)
        self.params = dict(DEFAULT_PARAMS, **(params or {}))
        for k in _UNSUPPORTED:
            if self.flags.get(k):
                raise NotImplementedError(f"augmentation '{k}' is not part of the GPU recipe "
                                          "(rotate, crop, random_crop, resize, color_jitter)")
        if not self.flags.get("resize"):
            raise NotImplementedError("the GPU augmenter needs resize=True: crops have per-sample sizes otherwise")
        self.rng = rng if rng is not None else _random
        self.channels_last = channels_last

    # ---- host: parameters of one view (sample_augmenter.py:47-129, parameter side)
    def _crop_size(self, joints: Tensor, jitter: Sequence[int], crop_margin: float) -> Tuple[int, int, int, int, int]:
        center_y, center_x = int(torch.mean(joints[:, 1])), int(torch.mean(joints[:, 0]))
        far = torch.max((joints[:, 1] - center_y) ** 2 + (joints[:, 0] - center_x) ** 2)
        side = int(far ** 0.5 * crop_margin)
        origin_x = max(center_x - side + jitter[0], 0)
        origin_y = max(center_y - side + jitter[1], 0)
        return origin_x, origin_y, int(2 * side), center_x - side - origin_x, center_y - side - origin_y

    def sample_view(self, joints25d: Tensor, image_hw: Tuple[int, int]) -> Dict:
        f, p, rng = self.flags, self.params, self.rng
        h_img, w_img = image_hw
        joints = joints25d.detach().to("cpu", torch.float32).clone()
        view: Dict = {"angle": None, "h": None, "s": None, "a": None, "b": None, "blur_flag": False, "minv": None}
        if f.get("rotate"):
            ox, oy, side, _, _ = self._crop_size(joints, (0, 0), 0.0)
            center = (int(ox + side / 2), int(oy + side / 2))
            angle = rng.uniform(p["max_angle"], p["min_angle"]) // 1  # the reference swaps min/max on load
            rot = _rotation_matrix(center, angle)
            hom = joints.double()
            hom[:, -1] = 1.0
            joints[:, :-1] = (hom @ torch.tensor(rot, dtype=torch.float64).T).float()
            view["angle"], view["minv"] = angle, _invert_affine(rot)
        # hybrid2 always crops: with the crop flag off it passes a zero jitter (data_set.py:359-364)
        if f.get("random_crop"):
            margin = rng.uniform(p["crop_margin_range"][0], p["crop_margin_range"][1])
        else:
            margin = p["crop_margin"]
        if f.get("crop"):
            jitter = (int(rng.uniform(0, p["crop_box_jitter"][1])), int(rng.uniform(0, p["crop_box_jitter"][1])))
        else:
            jitter = (0, 0)
        ox, oy, side, jx, jy = self._crop_size(joints, jitter, margin)
        x0, y0 = min(ox, w_img), min(oy, h_img)
        cw, ch = min(ox + side, w_img) - x0, min(oy + side, h_img) - y0
        if cw <= 0 or ch <= 0:
            raise ValueError(f"empty crop window (origin {ox},{oy}, side {side}) for a {w_img}x{h_img} image: "
                             "the reference's cv2.resize fails on it too")
        view.update(jitter_x=jx, jitter_y=jy, crop_margin_scale=margin, crop=(x0, y0, cw, ch))
        if f.get("color_jitter"):
            view["h"] = rng.uniform(*p["hue_factor_range"])
            view["s"] = rng.uniform(*p["sat_factor_range"])
            view["a"] = rng.uniform(*p["value_factor_alpha_range"])
            view["b"] = rng.uniform(*p["value_factor_beta_range"])
        return view

    @staticmethod
    def pack(view: Dict) -> List[float]:
        """One record of include/peclr_hip.h's `params` layout."""
        rot = view["minv"] is not None
        col = view["h"] is not None
        return ([*(view["minv"] if rot else [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]), float(rot), *map(float, view["crop"]),
                 float(col), *((view["h"], view["s"], view["a"], view["b"]) if col else (1.0, 1.0, 1.0, 0.0))])

    def sample_batch(self, joints25d: Tensor, image_hw: Tuple[int, int]):
        """Draws view 1 then view 2 for each sample in turn (the order a dataset iterates)."""
        views: List[List[Dict]] = [[], []]
        for j in joints25d:
            for v in (0, 1):
                views[v].append(self.sample_view(j, image_hw))
        params = torch.tensor([[self.pack(w) for w in views[v]] for v in (0, 1)], dtype=torch.float64)
        return params, views

    @staticmethod
    def collate(views: List[List[Dict]]) -> Dict[str, Tensor]:
        out: Dict[str, Tensor] = {}
        for v in (0, 1):
            for key in ("angle", "jitter_x", "jitter_y", "h", "s", "a", "b", "blur_flag", "crop_margin_scale"):
                vals = [w[key] for w in views[v]]
                if vals[0] is None:
                    continue  # the reference drops None entries (data_set.py:382-383)
                if isinstance(vals[0], bool):
                    t = torch.tensor(vals, dtype=torch.bool)
                elif isinstance(vals[0], int):
                    t = torch.tensor(vals, dtype=torch.int64)
                else:
                    t = torch.tensor(vals, dtype=torch.float64)
                out[f"{key}_{v + 1}"] = t
        return out

    # ---- device: the batch
    def __call__(self, images: Tensor, joints25d: Tensor) -> Dict[str, Tensor]:
        """images: [B,H,W,3] uint8 on the HIP device (the reference's RGB HWC arrays, one size per
        batch); joints25d: [B,21,3] (any device; the parameter logic runs on the host)."""
        b, h, w, _ = images.shape
        params, views = self.sample_batch(joints25d, (h, w))
        rw, rh = self.params["resize_shape"]
        out, _ = _capi.augment_views(images, params.to(images.device, non_blocking=True), (rh, rw), IMAGENET_MEAN,
                                     IMAGENET_STD, self.channels_last)
        # both views are the halves of ONE buffer; `transformed_images` lets the model skip its cat(view1, view2)
        # (hybrid2_model.py:30-32) -- an extra "image" entry, which the reference's consumers of the dict ignore
        batch = {"transformed_images": out, "transformed_image1": out[:b], "transformed_image2": out[b:]}
        batch.update({k: t.to(images.device, non_blocking=True) for k, t in self.collate(views).items()})
        return batch
