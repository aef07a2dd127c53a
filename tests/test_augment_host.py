"""Two-view augmentation, CPU side: the oracle's and the product's PARAMETER logic against vectors
captured from the reference (tests/golden/g9_augment_params.json, made by make_golden_augment.py),
and known-answer checks of the oracle's pixel restatement (which is unpinned: no OpenCV here)."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment_oracle as A
from tests.conftest import GOLDEN as GOLDEN_DIR

ALL_FLAGS = ["color_drop", "color_jitter", "crop", "cut_out", "gaussian_blur", "random_crop", "resize", "rotate",
             "gaussian_noise", "sobel_filter"]


def cases():
    with open(os.path.join(GOLDEN_DIR, "g9_augment_params.json")) as f:
        return json.load(f)["cases"]


CASES = cases()
IDS = [c["name"] for c in CASES]


def flags_of(c):
    return {k: k in c["flags_on"] for k in ALL_FLAGS}


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_parameters_equal_reference(case):
    flags = flags_of(case)
    j25 = A.convert_to_2_5d(np.array(case["K"]), np.array(case["joints3D"]))
    rng = random.Random(case["seed"])
    override = None if flags["crop"] else [0, 0]
    for i, gv in enumerate(case["views"]):
        v = A.sample_view(j25, tuple(case["image_hw"]), flags, case["params"], rng, override)
        for k in ("angle", "jitter_x", "jitter_y", "h", "s", "a", "b", "crop_margin_scale"):
            key = f"{k}_{i + 1}"
            if key in case["emitted"]:
                assert float(v[k]) == case["emitted"][key]["value"], key      # exact: same draws, same arithmetic
            else:
                assert v[k] is None, key
        assert [v["box"]["origin_x"], v["box"]["origin_y"], v["box"]["side"]] == gv["boxes"][-1]
        np.testing.assert_array_equal(v["joints"].astype(np.float64), np.array(gv["joints"]))
        np.testing.assert_allclose(v["T"], np.array(gv["T"]), rtol=0, atol=1e-12)
        calls = {c[0]: c for c in gv["calls"]}
        if "warpAffine" in calls:
            np.testing.assert_array_equal(v["rot"], np.array(calls["warpAffine"][1]))
        assert calls["resize"][1] == [v["crop"][3], v["crop"][2]]            # the window the reference resizes


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_product_parameters_equal_reference(case):
    """peclr_amd.augment draws the same parameters, in the same order, and emits the same dict types."""
    from peclr_amd.augment import TwoViewAugmenter, convert_to_2_5d

    flags = flags_of(case)
    aug = TwoViewAugmenter(flags, case["params"], rng=random.Random(case["seed"]))
    j25, _ = convert_to_2_5d(torch.tensor(case["K"], dtype=torch.float32), torch.tensor(case["joints3D"], dtype=torch.float32))
    params, views = aug.sample_batch(j25[None], tuple(case["image_hw"]))
    emitted = aug.collate(views)
    expect_dtype = {"float": torch.float64, "int": torch.int64, "bool": torch.bool}
    assert set(emitted) == set(case["emitted"])
    for key, ref in case["emitted"].items():
        assert emitted[key].dtype == expect_dtype[ref["type"]], key
        assert float(emitted[key][0]) == ref["value"], key
    for i, gv in enumerate(case["views"]):
        w = views[i][0]
        box = gv["boxes"][-1]
        h_img, w_img = case["image_hw"]
        x0, y0 = min(box[0], w_img), min(box[1], h_img)
        assert w["crop"] == (x0, y0, min(box[0] + box[2], w_img) - x0, min(box[1] + box[2], h_img) - y0)
        calls = {c[0]: c for c in gv["calls"]}
        if "warpAffine" in calls:
            np.testing.assert_allclose(np.array(w["minv"]).reshape(2, 3), A.invert_affine(np.array(calls["warpAffine"][1])),
                                       rtol=0, atol=1e-15)
        rec = params[i, 0].tolist()
        assert rec[6] == float("warpAffine" in calls) and rec[7:11] == [float(t) for t in w["crop"]]


def test_product_rejects_flags_outside_the_recipe():
    from peclr_amd.augment import TwoViewAugmenter

    with pytest.raises(NotImplementedError, match="gaussian_blur"):
        TwoViewAugmenter({"resize": True, "gaussian_blur": True})
    with pytest.raises(NotImplementedError, match="resize"):
        TwoViewAugmenter({"crop": True})
    aug = TwoViewAugmenter()
    with pytest.raises(Exception, match="HIP tensor|no CPU path"):
        aug(torch.zeros(1, 32, 32, 3, dtype=torch.uint8), torch.rand(1, 21, 3) * 20)


# ------------------------------------------------------------------ oracle pixel restatement: known answers
def test_warp_identity_quarter_turn_and_window():
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (96, 120, 3), dtype=np.uint8)
    assert np.array_equal(A.warp_affine_u8(img, A.rotation_matrix_2d((40, 50), 0.0)), img)
    sq = g.integers(0, 256, (101, 101, 3), dtype=np.uint8)
    turned = A.warp_affine_u8(sq, A.rotation_matrix_2d((50, 50), 90.0))      # centre pixel: exact permutation
    assert np.array_equal(turned, np.rot90(sq, 1))
    m = A.rotation_matrix_2d((61, 47), 33.0)
    full = A.warp_affine_u8(img, m)
    assert np.array_equal(A.warp_affine_u8(img, m, region=(30, 20, 70, 50)), full[20:70, 30:100])
    black = A.warp_affine_u8(img, A.rotation_matrix_2d((5000, 5000), 45.0))
    assert black.max() == 0                                                  # everything maps outside the source


@pytest.mark.parametrize("src_wh,mode", [((256, 256), "area_fast"), ((384, 256), "area_fast"), ((152, 152), "area"),
                                         ((200, 170), "area"), ((129, 128), "area"), ((100, 100), "linear"),
                                         ((90, 140), "linear"), ((128, 128), "copy")])
def test_resize_paths(src_wh, mode):
    sw, sh = src_wh
    assert A.resize_mode(sw, sh, 128, 128) == mode
    flat = A.resize_area_u8(np.full((sh, sw, 3), 77, np.uint8), (128, 128))
    assert flat.shape == (128, 128, 3) and (flat == 77).all()                # weights sum to one on every path
    g = np.random.default_rng(sw + sh)
    src = g.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    out = A.resize_area_u8(src, (128, 128))
    assert abs(out.mean() - src.mean()) < 0.6
    if mode == "area_fast" and (sw, sh) == (256, 256):
        blk = src.astype(np.int64).reshape(128, 2, 128, 2, 3).sum(axis=(1, 3))
        assert np.array_equal(out, ((blk + 2) >> 2).astype(np.uint8))
    ramp = np.repeat(np.linspace(0, 255, sw).astype(np.uint8)[None, :, None], sh, 0).repeat(3, 2)
    r = A.resize_area_u8(ramp, (128, 128)).astype(int)
    assert (np.diff(r[0, :, 0]) >= 0).all()                                  # monotone input stays monotone


def test_hsv_known_answers_and_jitter():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 64, 32]]], np.uint8)
    # channel 0 is "blue": H = 120 (of 180), green 60, red 0; greys have S = 0
    assert A.bgr2hsv_u8(px)[0].tolist() == [[120, 255, 255], [60, 255, 255], [0, 255, 255], [0, 0, 255], [0, 0, 0],
                                             [110, 191, 128]]
    assert A.hsv2bgr_u8(A.bgr2hsv_u8(px))[0, :5].tolist() == px[0, :5].tolist()
    g = np.random.default_rng(3)
    img = g.integers(0, 256, (40, 40, 3), dtype=np.uint8)
    back = A.hsv2bgr_u8(A.bgr2hsv_u8(img))
    assert np.abs(back.astype(int) - img.astype(int)).max() <= 6            # 8-bit HSV quantisation
    dark = A.color_jitter_u8(img, 1.0, 1.0, 0.5, 0.0)
    assert dark.max(axis=2).astype(int).max() <= 128                         # V halves
    grey = A.color_jitter_u8(img, 1.0, 0.0, 1.0, 0.0)
    assert (grey.max(axis=2) == grey.min(axis=2)).all()                      # S = 0 -> achromatic
    assert np.array_equal(grey.max(axis=2), img.max(axis=2))                 # ... with V kept


def test_to_tensor_normalize_matches_torch_ops():
    g = np.random.default_rng(4)
    img = g.integers(0, 256, (8, 9, 3), dtype=np.uint8)
    t = torch.from_numpy(img.transpose(2, 0, 1).copy()).float().div(255)
    mean, std = torch.tensor(A.IMAGENET_MEAN), torch.tensor(A.IMAGENET_STD)
    ref = t.sub(mean[:, None, None]).div(std[:, None, None])                 # what ToTensor + Normalize execute
    assert np.array_equal(A.to_tensor_normalize(img), ref.numpy())


def test_prepare_hybrid2_sample_dict():
    c = CASES[0]
    flags = flags_of(c)
    j25 = A.convert_to_2_5d(np.array(c["K"]), np.array(c["joints3D"]))
    img = np.random.default_rng(1).integers(0, 256, (*c["image_hw"], 3), dtype=np.uint8)
    out = A.prepare_hybrid2_sample(img, j25, flags, c["params"], random.Random(c["seed"]))
    assert out["transformed_image1"].shape == (3, 128, 128) and out["transformed_image1"].dtype == np.float32
    for k, ref in c["emitted"].items():
        assert float(out[k]) == ref["value"], k
    assert not np.array_equal(out["transformed_image1"], out["transformed_image2"])


# ---- pixel half: cross-checks against INDEPENDENT implementations of the same published operations.
# OpenCV itself is not installed (the restatement of its 8-bit fixed-point arithmetic stays unpinned), but the
# operations are standard: an affine warp with bilinear interpolation (SciPy), area-averaging resize (exact
# integration of the source over each destination cell) and RGB<->HSV (the standard library).  OpenCV's 8-bit
# paths differ from the exact operations only by their fixed-point rounding, so the restatement must agree with
# these references to within 1-2 grey levels on smooth content; a wrong centre, axis order, interpolation weight or
# hue sector would be off by tens.
def _smooth_image(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 90 * np.sin(xx / rng.uniform(9, 25) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(9, 25)),
                    127 + 100 * np.cos((xx + yy) / rng.uniform(12, 30) + rng.uniform(0, 6)),
                    127 + 80 * np.sin(yy / rng.uniform(8, 20) + rng.uniform(0, 6))], axis=2)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("angle", [0.0, 17.0, -33.5, 45.0, 90.0])
def test_warp_affine_agrees_with_scipy_bilinear(angle):
    from scipy import ndimage

    from oracle import augment_oracle as A

    img = _smooth_image(96, 112, 3)
    m = A.rotation_matrix_2d((56, 48), angle)                     # forward matrix, as cv2.getRotationMatrix2D
    got = A.warp_affine_u8(img, m).astype(np.float64)
    mi = A.invert_affine(m)                                       # dst (x, y) -> src (x, y)
    # scipy works in (row, col) = (y, x): src_rc = M_rc @ dst_rc + off_rc
    m_rc = np.array([[mi[1, 1], mi[1, 0]], [mi[0, 1], mi[0, 0]]])
    off = np.array([mi[1, 2], mi[0, 2]])
    want = np.stack([ndimage.affine_transform(img[..., c].astype(np.float64), m_rc, offset=off, order=1, mode="constant", cval=0.0)
                     for c in range(3)], axis=2)
    inner = np.zeros(img.shape[:2], bool)
    inner[2:-2, 2:-2] = True
    # where the source footprint lies fully inside the image both are plain bilinear interpolation
    src_x = mi[0, 0] * np.arange(112)[None, :] + mi[0, 1] * np.arange(96)[:, None] + mi[0, 2]
    src_y = mi[1, 0] * np.arange(112)[None, :] + mi[1, 1] * np.arange(96)[:, None] + mi[1, 2]
    inside = inner & (src_x > 1) & (src_x < 110) & (src_y > 1) & (src_y < 94)
    diff = np.abs(got - want)[inside]
    assert inside.sum() > 3000 and diff.max() <= 2.0 and (diff > 1.0).mean() < 0.02, (diff.max(), (diff > 1.0).mean())


@pytest.mark.parametrize("src_hw,dst_wh", [((120, 90), (40, 30)), ((97, 131), (64, 64)), ((100, 100), (37, 53)), ((64, 64), (128, 128))])
def test_resize_area_agrees_with_exact_area_integration(src_hw, dst_wh):
    """INTER_AREA when shrinking = the mean of the source over each destination cell (cells have fractional edges);
    when enlarging OpenCV switches to bilinear with area-mode coefficients: checked against the exact cell mean only
    for shrinking, and for monotonicity / range when enlarging."""
    from oracle import augment_oracle as A

    sh, sw = src_hw
    dw, dh = dst_wh
    img = _smooth_image(sh, sw, 11)
    got = A.resize_area_u8(img, (dw, dh)).astype(np.float64)
    assert got.shape == (dh, dw, 3)
    if dw > sw or dh > sh:
        assert got.min() >= img.min() - 1 and got.max() <= img.max() + 1
        return

    def weights(s, d):                      # [d, s] overlap of source pixel [j, j+1) with destination cell [i*s/d, (i+1)*s/d)
        scale = s / d
        wm = np.zeros((d, s))
        for i in range(d):
            lo, hi = i * scale, (i + 1) * scale
            for j in range(int(np.floor(lo)), min(s, int(np.ceil(hi)))):
                wm[i, j] = max(0.0, min(hi, j + 1) - max(lo, j))
        return wm / scale

    wy, wx = weights(sh, dh), weights(sw, dw)
    want = np.einsum("ij,jkc,lk->ilc", wy, img.astype(np.float64), wx)
    diff = np.abs(got - want)
    assert diff.max() <= 1.0 and (diff > 0.51).mean() < 0.01, (diff.max(), (diff > 0.51).mean())


def test_hsv_round_trip_agrees_with_colorsys():
    import colorsys

    from oracle import augment_oracle as A

    rng = np.random.default_rng(5)
    px = rng.integers(0, 256, (4000, 3), dtype=np.uint8)
    px[:6] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128]]
    img = px.reshape(40, 100, 3)
    hsv = A.bgr2hsv_u8(img).reshape(-1, 3).astype(np.float64)
    # OpenCV 8-bit HSV: channel order (b, g, r) in, H in [0, 180) = degrees / 2, S and V scaled to 255
    ref = np.array([colorsys.rgb_to_hsv(r / 255.0, g / 255.0, b / 255.0) for b, g, r in px.astype(np.float64)])
    ref_h, ref_s, ref_v = ref[:, 0] * 180.0, ref[:, 1] * 255.0, ref[:, 2] * 255.0
    dh = np.abs(hsv[:, 0] - ref_h)
    dh = np.minimum(dh, 180.0 - dh)                                # hue is circular
    grey = ref_s < 1.0                                              # hue of a grey pixel is arbitrary
    assert dh[~grey].max() <= 1.0 and np.abs(hsv[:, 1] - ref_s).max() <= 1.0 and np.abs(hsv[:, 2] - ref_v).max() == 0.0
    back = A.hsv2bgr_u8(A.bgr2hsv_u8(img)).astype(np.int64)
    # quantising H to 180 steps and S to 8 bits loses information; the round trip stays within a few levels
    assert np.abs(back - img.astype(np.int64)).max() <= 6 and np.abs(back - img.astype(np.int64)).mean() < 1.2
