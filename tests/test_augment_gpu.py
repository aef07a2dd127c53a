"""Two-view augmentation on the GPU: csrc/augment.hip through the C ABI against the NumPy restatement
(oracle/augment_oracle.py) -- bit-exact, 8-bit stage by 8-bit stage -- and the emitted dict driving a
real Hybrid2Model step."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment_oracle as A
from tests.conftest import GOLDEN as GOLDEN_DIR

pytestmark = pytest.mark.gpu
DEV = "cuda"
ALL_FLAGS = ["color_drop", "color_jitter", "crop", "cut_out", "gaussian_blur", "random_crop", "resize", "rotate",
             "gaussian_noise", "sobel_filter"]

with open(os.path.join(GOLDEN_DIR, "g9_augment_params.json")) as f:
    CASES = json.load(f)["cases"]


def synth_image(seed, hw):
    """Smooth structure + texture, so interpolation errors would show (pure noise hides them)."""
    g = np.random.default_rng(seed)
    h, w = hw
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 120 * np.sin(xx / 17.0 + seed), 127 + 120 * np.cos(yy / 11.0), 60 + (xx + yy) % 190], axis=2)
    return np.clip(base + g.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_kernels_equal_oracle_on_reference_parameter_sets(case):
    from peclr_amd import _capi
    from peclr_amd.augment import IMAGENET_MEAN, IMAGENET_STD, TwoViewAugmenter, convert_to_2_5d

    flags = {k: k in case["flags_on"] for k in ALL_FLAGS}
    aug = TwoViewAugmenter(flags, case["params"], rng=random.Random(case["seed"]), channels_last=True)
    hw = tuple(case["image_hw"])
    image = synth_image(case["seed"], hw)
    j25, _ = convert_to_2_5d(torch.tensor(case["K"], dtype=torch.float32), torch.tensor(case["joints3D"], dtype=torch.float32))
    params, views = aug.sample_batch(j25[None], hw)
    rw, rh = case["params"]["resize_shape"]
    out, crops = _capi.augment_views(torch.from_numpy(image)[None].to(DEV), params.to(DEV), (rh, rw), IMAGENET_MEAN,
                                     IMAGENET_STD, channels_last=True)
    assert out.shape == (2, 3, rh, rw) and out.is_contiguous(memory_format=torch.channels_last)
    for v in (0, 1):
        w = views[v][0]
        ov = {"rot": None if w["minv"] is None else np.array(w["minv"]).reshape(2, 3), "crop": w["crop"], "resized": True,
              "h": w["h"], "s": w["s"], "a": w["a"], "b": w["b"]}
        x0, y0, cw, ch = w["crop"]
        # stage 1: the crop window of the rotated image
        win = crops[v, 0, :ch, :cw].cpu().numpy()
        if ov["rot"] is None:
            assert np.array_equal(win, image[y0:y0 + ch, x0:x0 + cw])
        else:
            assert np.array_equal(win, warp_with_inverse(image, ov["rot"], (x0, y0, cw, ch)))
        # stage 2 on the oracle side, from the same 8-bit window
        img = A.resize_area_u8(win, (rw, rh))
        if flags["color_jitter"]:
            img = A.color_jitter_u8(img, w["h"], w["s"], w["a"], w["b"])
        ref = A.to_tensor_normalize(img)
        got = out[v].cpu().numpy()
        assert np.array_equal(got, ref), f"view {v}: max |d| = {np.abs(got - ref).max()}"


def warp_with_inverse(image, minv, region):
    """oracle warp driven by the already-inverted matrix (what the product hands to the kernel)."""
    orig = A.invert_affine
    try:
        A.invert_affine = lambda m: np.asarray(minv, dtype=np.float64)
        return A.warp_affine_u8(image, np.eye(2, 3), region=region)
    finally:
        A.invert_affine = orig


@pytest.mark.parametrize("src_wh", [(256, 256), (152, 152), (200, 170), (100, 100), (90, 140), (128, 128), (224, 224),
                                    (129, 128), (64, 64), (127, 300), (300, 127)])
@pytest.mark.parametrize("nhwc", [True, False], ids=["nhwc", "nchw"])
def test_every_resize_path_bit_exact(src_wh, nhwc):
    from peclr_amd import _capi
    from peclr_amd.augment import IMAGENET_MEAN, IMAGENET_STD

    sw, sh = src_wh
    h, w = 320, 320
    image = synth_image(sw * 7 + sh, (h, w))
    rec = [1.0, 0, 0, 0, 1.0, 0, 0.0, 3.0, 5.0, float(sw), float(sh), 1.0, 0.73, 0.44, 0.9, 13.0]
    params = torch.tensor([[rec]], dtype=torch.float64)
    out, crops = _capi.augment_views(torch.from_numpy(image)[None].to(DEV), params.to(DEV), (128, 128), IMAGENET_MEAN,
                                     IMAGENET_STD, channels_last=nhwc)
    win = image[5:5 + sh, 3:3 + sw]
    assert np.array_equal(crops[0, 0, :sh, :sw].cpu().numpy(), win)
    img = A.color_jitter_u8(A.resize_area_u8(win, (128, 128)), 0.73, 0.44, 0.9, 13.0)
    assert np.array_equal(out[0].cpu().numpy(), A.to_tensor_normalize(img)), A.resize_mode(sw, sh, 128, 128)


def test_rotation_sweep_bit_exact():
    from peclr_amd import _capi
    from peclr_amd.augment import IMAGENET_MEAN, IMAGENET_STD, _invert_affine, _rotation_matrix

    image = synth_image(5, (224, 224))
    recs, minvs = [], []
    for angle in range(-45, 46, 5):
        minv = _invert_affine(_rotation_matrix((100 + angle // 9, 120 - angle // 7), float(angle)))
        minvs.append(minv)
        recs.append([*minv, 1.0, 0.0, 0.0, 224.0, 224.0, 0.0, 1.0, 1.0, 1.0, 0.0])
    b = len(recs)
    params = torch.tensor([recs], dtype=torch.float64)
    images = torch.from_numpy(image)[None].expand(b, -1, -1, -1).contiguous().to(DEV)
    out, crops = _capi.augment_views(images, params.to(DEV), (224, 224), IMAGENET_MEAN, IMAGENET_STD, channels_last=False)
    for i in range(b):
        ref = warp_with_inverse(image, np.array(minvs[i]).reshape(2, 3), (0, 0, 224, 224))
        assert np.array_equal(crops[0, i].cpu().numpy(), ref), f"angle index {i}"
        assert np.array_equal(out[i].cpu().numpy(), A.to_tensor_normalize(ref))     # resize == copy, no jitter


def test_batch_dict_drives_a_training_step():
    """TwoViewAugmenter -> batch dict -> Hybrid2Model.training_step on the HIP kernels; both views come out
    of one launch pair, dtypes are the reference's collate dtypes."""
    import warnings

    from peclr_amd import Hybrid2Model, Trainer, TwoViewAugmenter, hybrid2_config
    from peclr_amd.bn2d import enable_hip_batchnorm

    warnings.simplefilter("ignore")
    b = 8
    g = np.random.default_rng(2)
    images = np.stack([synth_image(i, (224, 224)) for i in range(b)])
    joints = torch.from_numpy(np.concatenate([g.normal((112, 108), 25, (b, 21, 2)), g.normal(0, 1, (b, 21, 1))], axis=2)).float()
    aug = TwoViewAugmenter(params={"resize_shape": [64, 64]}, rng=random.Random(3))
    batch = aug(torch.from_numpy(images).to(DEV), joints)
    assert batch["transformed_image1"].shape == (b, 3, 64, 64) and batch["transformed_image1"].dtype == torch.float32
    assert batch["angle_1"].dtype == torch.float64 and batch["jitter_x_2"].dtype == torch.int64
    assert batch["blur_flag_1"].dtype == torch.bool and batch["h_1"].dtype == torch.float64
    assert all(t.is_cuda for t in batch.values())
    # the same draws through the oracle give the same tensors
    rng = random.Random(3)
    ref = [A.prepare_hybrid2_sample(images[i], joints[i].numpy(), aug.flags, aug.params, rng) for i in range(b)]
    for i in range(b):
        assert np.array_equal(batch["transformed_image1"][i].cpu().numpy(), ref[i]["transformed_image1"])
        assert np.array_equal(batch["transformed_image2"][i].cpu().numpy(), ref[i]["transformed_image2"])
        assert float(batch["angle_2"][i]) == ref[i]["angle_2"] and int(batch["jitter_y_1"][i]) == ref[i]["jitter_y_1"]
    torch.manual_seed(0)
    cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, augmentation=["crop", "rotate"], batch_size=b,
                         num_samples=64, pretrained=False)
    model = Hybrid2Model(cfg).to(DEV).train()
    model.encoder = model.encoder.to(memory_format=torch.channels_last)
    enable_hip_batchnorm(model.encoder)
    trainer = Trainer(max_epochs=1).attach(model)
    out = trainer.training_micro_step(batch, 0)
    assert torch.isfinite(out["loss"]).item() and len(out) == 17
