"""Golden vectors for the two-view augmentation's PARAMETER logic.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_augment.py

What the reference computes itself -- the order of `random.uniform` draws, the crop box
(`SampleAugmenter.get_crop_size`, sample_augmenter.py:434-478), the rotation centre
(`rotate_sample`, :218-247), `jitter_x/jitter_y`, the rotated joints, the 3x3 transformation
matrix and the dict `Data_Set.prepare_hybrid2_sample` emits (data_set.py:357-384) -- is captured
here by calling the reference's own code.

What it delegates to OpenCV is NOT installed in this image, so the pixel operations are replaced
by recording pass-throughs (`warpAffine` returns its input, `resize` returns zeros of the requested
shape, `cvtColor`/`split`/`merge` are shape-preserving): the fixtures therefore pin the host-side
parameter logic only; the pixel arithmetic stays "parity unpinned" (oracle/augment_oracle.py).
`cv2.getRotationMatrix2D` is supplied from OpenCV's documented formula because the reference's
joint rotation needs its value.

Fixture: g9_augment_params.json -- per case: inputs (flags, params, seed, K, joints3D, image
shape) and, per view, everything listed above.
"""
import json
import math
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import install_stubs  # noqa: E402

edict = install_stubs()
_tt = types.ModuleType("torch.tensor")  # `from torch.tensor import Tensor` (torch 1.7 layout)
_tt.Tensor = torch.Tensor
sys.modules["torch.tensor"] = _tt

cv2 = sys.modules["cv2"]
CALLS = []


def _get_rotation_matrix_2d(center, angle, scale):
    a = angle * math.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    return np.array([[alpha, beta, (1 - alpha) * center[0] - beta * center[1]],
                     [-beta, alpha, beta * center[0] + (1 - alpha) * center[1]]], dtype=np.float64)


def _warp_affine(image, m, dsize):
    CALLS.append(("warpAffine", np.asarray(m, dtype=np.float64).tolist(), list(dsize)))
    return image


def _resize(image, dsize, interpolation=None):
    CALLS.append(("resize", list(image.shape[:2]), list(dsize)))
    if image.shape[0] == 0 or image.shape[1] == 0:
        raise ValueError("empty source")  # OpenCV raises on an empty source; the reference catches it
    return np.zeros((dsize[1], dsize[0], image.shape[2]), dtype=image.dtype)


cv2.getRotationMatrix2D = _get_rotation_matrix_2d
cv2.warpAffine = _warp_affine
cv2.resize = _resize
cv2.INTER_AREA = 3
cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR = 40, 54
cv2.cvtColor = lambda img, code: img
cv2.split = lambda img: [img[:, :, i] for i in range(img.shape[2])]
cv2.merge = lambda chans: np.stack(chans, axis=2)

from src.data_loader.data_set import Data_Set  # noqa: E402
from src.data_loader.sample_augmenter import SampleAugmenter  # noqa: E402

PARAMS = {"crop_margin": 1.25, "crop_margin_range": [0.9, 1.5], "cut_out_fraction": [0.0, 0.16],
          "hue_factor_range": [0.01, 1.0], "max_angle": 45, "min_angle": -45, "resize_shape": [128, 128],
          "sat_factor_range": [0.01, 1.0], "value_factor_alpha_range": [0.5, 1], "value_factor_beta_range": [5, 20],
          "crop_box_jitter": [0.0, 15.0], "sobel_kernel": 3, "noise_std": 25}
ALL_FLAGS = ["color_drop", "color_jitter", "crop", "cut_out", "gaussian_blur", "random_crop", "resize", "rotate",
             "gaussian_noise", "sobel_filter"]


def flags(*on):
    return {k: (k in on) for k in ALL_FLAGS}


def make_sample(seed, hw, centre, spread, depth=0.6):
    """A synthetic hand: 21 3-D joints whose projection lands around `centre` (pixels)."""
    g = np.random.default_rng(seed)
    k = np.array([[480.0, 0, hw[1] / 2], [0, 480.0, hw[0] / 2], [0, 0, 1]], dtype=np.float32)
    z = depth + 0.05 * g.standard_normal(21)
    u = centre[0] + spread * g.standard_normal(21)
    v = centre[1] + spread * g.standard_normal(21)
    j3 = np.stack([(u - k[0, 2]) * z / k[0, 0], (v - k[1, 2]) * z / k[1, 1], z], axis=1).astype(np.float32)
    image = g.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    return {"image": image, "K": torch.from_numpy(k), "joints3D": torch.from_numpy(j3)}


def run_case(name, on, seed, hw, centre, spread, params=None):
    p = dict(PARAMS, **(params or {}))
    aug = SampleAugmenter(edict(flags(*on)), edict(p))
    boxes = []
    orig = aug.get_crop_size

    def spy(joints, jitter=None, crop_margin=None):
        out = orig(joints, jitter, crop_margin)
        boxes.append([int(v) for v in out])
        return out

    aug.get_crop_size = spy
    views = []
    orig_transform = aug.transform_sample

    def transform_spy(image, joints, override_angle=None, override_jitter=None):
        CALLS.clear()
        boxes.clear()
        img, joints_out, t = orig_transform(image, joints, override_angle, override_jitter)
        views.append({"calls": [list(c) for c in CALLS], "boxes": [list(b) for b in boxes],
                      "out_shape": list(img.shape), "joints": joints_out.double().numpy().tolist(),
                      "T": np.asarray(t, dtype=np.float64).tolist()})
        return img, joints_out, t

    aug.transform_sample = transform_spy
    sample = make_sample(seed, hw, centre, spread)
    fake = types.SimpleNamespace(transform=None)
    fake.get_random_augment_param = lambda a: Data_Set.get_random_augment_param(fake, a)
    random.seed(seed)
    out = Data_Set.prepare_hybrid2_sample(fake, sample, aug)
    emitted = {}
    for k, v in out.items():
        if k.startswith("transformed_image"):
            continue
        emitted[k] = {"type": type(v).__name__, "value": (bool(v) if isinstance(v, bool) else float(v))}
    return {"name": name, "flags_on": list(on), "params": p, "seed": seed, "image_hw": list(hw),
            "K": sample["K"].numpy().tolist(), "joints3D": sample["joints3D"].double().numpy().tolist(),
            "views": views, "emitted": emitted}


def main():
    torch.set_num_threads(1)
    recipe = ("color_jitter", "random_crop", "rotate", "crop", "resize")  # the published PeCLR recipe (README.md)
    cases = []
    for seed in range(6):
        cases.append(run_case(f"recipe_{seed}", recipe, 100 + seed, (224, 224), (112 + 9 * seed, 108 - 7 * seed), 24 + 3 * seed))
    cases.append(run_case("recipe_border_topleft", recipe, 7, (224, 224), (20, 14), 30))      # crop clamped at 0
    cases.append(run_case("recipe_border_bottomright", recipe, 8, (224, 224), (205, 214), 28))  # crop truncated by the slice
    cases.append(run_case("recipe_small_hand", recipe, 9, (224, 224), (120, 100), 9))          # crop < 128: upscaling
    cases.append(run_case("recipe_wide", recipe, 10, (240, 320), (170, 110), 35))
    cases.append(run_case("recipe_448", recipe, 11, (224, 224), (100, 120), 26, {"resize_shape": [448, 448]}))
    cases.append(run_case("crop_only", ("crop", "resize"), 12, (224, 224), (110, 115), 25))
    cases.append(run_case("rotate_only", ("rotate", "resize"), 13, (224, 224), (118, 104), 25))
    cases.append(run_case("resize_only", ("resize",), 14, (224, 224), (112, 112), 25))
    cases.append(run_case("color_only", ("color_jitter", "resize"), 15, (224, 224), (112, 112), 25))
    cases.append(run_case("fixed_margin", ("crop", "rotate", "resize", "color_jitter"), 16, (224, 224), (100, 130), 22))
    path = os.path.join(HERE, "g9_augment_params.json")
    with open(path, "w") as f:
        json.dump({"cases": cases}, f)
    print("wrote", path, os.path.getsize(path), "bytes;", len(cases), "cases")
    c = cases[0]
    print(json.dumps(c["emitted"], indent=0)[:600])
    print(c["views"][0]["boxes"], c["views"][0]["calls"])


if __name__ == "__main__":
    main()
