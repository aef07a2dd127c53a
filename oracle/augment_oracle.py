"""CPU restatement (NumPy) of the reference's two-view augmentation -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (peclr_amd/) never does.

Scope: the published PeCLR recipe (README.md: --color_jitter --random_crop --rotate --crop -resize)
as executed by `SampleAugmenter.transform_sample` (src/data_loader/sample_augmenter.py:47-129) and
`Data_Set.prepare_hybrid2_sample` (src/data_loader/data_set.py:357-384), followed by
ToTensor + Normalize (src/data_loader/utils.py:283-293).

Pinning status
  * PARAMETER logic (draw order of `random.uniform`, crop box, rotation centre/matrix, jitter_x/y,
    rotated joints, transformation matrix, emitted dict): PINNED by tests/golden/g9_augment_params.json,
    captured from the reference's own code (tests/golden/make_golden_augment.py).
  * PIXEL arithmetic: **parity unpinned**.  The reference delegates it to OpenCV
    (`opencv-python==4.4.0.46`, requirements.txt) which is not installed here and is not under
    /root/reference; the functions below restate OpenCV's published 8-bit algorithms
    (imgproc/src/imgwarp.cpp `warpAffine` fixed-point bilinear, imgproc/src/resize.cpp INTER_AREA
    incl. its integer fast path and the area-mode linear path used for up-scaling,
    imgproc/src/color_hsv.cpp 8-bit BGR<->HSV).  The reference has no test or fixture for them.
    Cross-checked (not pinned) against INDEPENDENT implementations of the same operations in
    tests/test_augment_host.py: scipy.ndimage.affine_transform (bilinear) for the warp, exact
    fractional-cell integration for INTER_AREA, colorsys for HSV -- agreement to <= 1-2 grey levels,
    which rules out geometric / sector / weight errors but not a different 8-bit rounding than OpenCV's.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

f32 = np.float32

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # data_loader/utils.py:288-290
IMAGENET_STD = (0.229, 0.224, 0.225)


# ------------------------------------------------------------------ parameter logic (pinned by g9)
def convert_to_2_5d(k: np.ndarray, joints3d: np.ndarray) -> np.ndarray:
    """data_loader/utils.py:19-33 (scale not needed on this path); float32 like the reference's tensors."""
    k, j = k.astype(f32), joints3d.astype(f32)
    parent, child = 0, 2  # wrist, index_mcp (data_loader/utils.py:15-16) -- only the depth column uses them
    out = ((k @ j.T).T / j[:, -1:]).astype(f32)
    scale = f32(np.sqrt(((j[child] - j[parent]) ** 2).sum(dtype=f32)))
    out[:, -1] = (j[:, -1] - j[parent, -1]) / scale
    return out


def rotation_matrix_2d(center: Tuple[int, int], angle: float) -> np.ndarray:
    """OpenCV getRotationMatrix2D(center, angle, 1.0) -- documented formula (sample_augmenter.py:432)."""
    a = angle * math.pi / 180.0
    al, be = math.cos(a), math.sin(a)
    return np.array([[al, be, (1 - al) * center[0] - be * center[1]],
                     [-be, al, be * center[0] + (1 - al) * center[1]]], dtype=np.float64)


def crop_size(joints: np.ndarray, jitter: Optional[Sequence[int]], crop_margin: float) -> Dict[str, int]:
    """sample_augmenter.py:434-478 with the margin and jitter already drawn.  float32 arithmetic as the
    reference's torch tensors; int() truncates toward zero."""
    j = joints.astype(f32)
    center_y = int(j[:, 1].sum(dtype=f32) / f32(j.shape[0]))
    center_x = int(j[:, 0].sum(dtype=f32) / f32(j.shape[0]))
    d2 = ((j[:, 1] - f32(center_y)) ** 2 + (j[:, 0] - f32(center_x)) ** 2).astype(f32)
    side = int(f32(np.sqrt(d2.max())) * f32(crop_margin))
    origin_x = max(center_x - side + jitter[0], 0)
    origin_y = max(center_y - side + jitter[1], 0)
    return {"origin_x": origin_x, "origin_y": origin_y, "side": int(2 * side),
            "jitter_x": center_x - side - origin_x, "jitter_y": center_y - side - origin_y}


def sample_view(joints25d: np.ndarray, image_hw: Tuple[int, int], flags: Dict[str, bool], params: Dict,
                rng, override_jitter=None) -> Dict:
    """One pass of transform_sample's parameter side (sample_augmenter.py:47-129) for the recipe's
    flags; `rng` is a `random.Random` (or the `random` module) -- the draw order is the reference's:
    angle, [crop margin], [jitter x, jitter y], [h, s, a, b]."""
    for k in ("sobel_filter", "cut_out", "gaussian_blur", "gaussian_noise", "color_drop"):
        if flags.get(k):
            raise NotImplementedError(f"augmentation '{k}' is outside the restated recipe")
    h_img, w_img = image_hw
    joints = joints25d.astype(f32).copy()
    t = np.identity(3)
    out: Dict = {"angle": None, "rot": None, "h": None, "s": None, "a": None, "b": None, "blur_flag": False}
    if flags.get("rotate"):
        box = crop_size(joints, [0, 0], 0.0)
        center = (int(box["origin_x"] + box["side"] / 2), int(box["origin_y"] + box["side"] / 2))
        # min_angle/max_angle are swapped by set_augmenation_params (sample_augmenter.py:486-487)
        angle = rng.uniform(params["max_angle"], params["min_angle"]) // 1
        rot = rotation_matrix_2d(center, angle)
        hom = joints.astype(np.float64).copy()
        hom[:, -1] = 1.0
        joints[:, :2] = (hom @ rot.T).astype(f32)
        t = np.concatenate((rot, np.array([[0, 0, 1.0]])))
        out.update(angle=angle, rot=rot)
    crop_box = None
    if flags.get("crop") or override_jitter is not None:
        if flags.get("random_crop"):
            margin = rng.uniform(params["crop_margin_range"][0], params["crop_margin_range"][1])
        else:
            margin = params["crop_margin"]
        jitter = override_jitter
        if jitter is None:
            jitter = [int(rng.uniform(0, params["crop_box_jitter"][1])), int(rng.uniform(0, params["crop_box_jitter"][1]))]
        box = crop_size(joints, jitter, margin)
        joints[:, 0] -= f32(box["origin_x"])
        joints[:, 1] -= f32(box["origin_y"])
        t[0, -1] -= box["origin_x"]
        t[1, -1] -= box["origin_y"]
        # numpy slicing clamps the box at the image border (sample_augmenter.py:183)
        x0, y0 = min(box["origin_x"], w_img), min(box["origin_y"], h_img)
        x1, y1 = min(box["origin_x"] + box["side"], w_img), min(box["origin_y"] + box["side"], h_img)
        crop_box = (x0, y0, x1 - x0, y1 - y0)
        out.update(jitter_x=box["jitter_x"], jitter_y=box["jitter_y"], crop_margin_scale=margin, box=box)
    else:
        out.update(jitter_x=None, jitter_y=None, crop_margin_scale=1.5, box=None)
    src_w, src_h = (crop_box[2], crop_box[3]) if crop_box else (w_img, h_img)
    out["resized"] = False
    if flags.get("resize"):
        rw, rh = params["resize_shape"]
        if src_w > 0 and src_h > 0:  # an empty crop makes cv2.resize raise; the reference prints and moves on
            fw, fh = rw / src_w, rh / src_h
            joints[:, 0] = joints[:, 0] * f32(fw)
            joints[:, 1] = joints[:, 1] * f32(fh)
            t[0] = t[0] * fw
            t[1] = t[1] * fh
            out["resized"] = True
    if flags.get("color_jitter"):
        out["h"] = rng.uniform(*params["hue_factor_range"])
        out["s"] = rng.uniform(*params["sat_factor_range"])
        out["a"] = rng.uniform(*params["value_factor_alpha_range"])
        out["b"] = rng.uniform(*params["value_factor_beta_range"])
    out.update(crop=crop_box, joints=joints, T=t)
    return out


# ------------------------------------------------------------------ pixel arithmetic (unpinned, see header)
def _rint(x):
    return np.rint(x).astype(np.int64)  # cvRound / saturate_cast<int>(double): round half to even


def invert_affine(m: np.ndarray) -> np.ndarray:
    """warpAffine inverts the forward matrix unless WARP_INVERSE_MAP (imgwarp.cpp)."""
    m = np.asarray(m, dtype=np.float64).copy()
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = a11, m[0, 1] * -d, m[1, 0] * -d, a22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    return m


def warp_affine_u8(img: np.ndarray, m_fwd: np.ndarray, region=None) -> np.ndarray:
    """8-bit warpAffine, INTER_LINEAR, BORDER_CONSTANT(0), dsize = source size: coordinates in 10-bit
    fixed point rounded to 1/32 pixel, bilinear weights as 15-bit integers.  region = (x0, y0, w, h)
    evaluates only that window of the destination."""
    h, w = img.shape[:2]
    mi = invert_affine(m_fwd)
    x0, y0, rw, rh = region if region is not None else (0, 0, w, h)
    xs = np.arange(x0, x0 + rw, dtype=np.float64)
    ys = np.arange(y0, y0 + rh, dtype=np.float64)
    adelta, bdelta = _rint(mi[0, 0] * xs * 1024.0), _rint(mi[1, 0] * xs * 1024.0)
    xrow = _rint((mi[0, 1] * ys + mi[0, 2]) * 1024.0) + 16
    yrow = _rint((mi[1, 1] * ys + mi[1, 2]) * 1024.0) + 16
    xf = (xrow[:, None] + adelta[None, :]) >> 5
    yf = (yrow[:, None] + bdelta[None, :]) >> 5
    sx, sy, fx, fy = xf >> 5, yf >> 5, xf & 31, yf & 31
    src = img.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok[..., None], src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0)

    w00, w01 = ((32 - fx) * (32 - fy) * 32)[..., None], (fx * (32 - fy) * 32)[..., None]
    w10, w11 = ((32 - fx) * fy * 32)[..., None], (fx * fy * 32)[..., None]
    acc = tap(sy, sx) * w00 + tap(sy, sx + 1) * w01 + tap(sy + 1, sx) * w10 + tap(sy + 1, sx + 1) * w11
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def _area_tab(ssize: int, dsize: int, scale: float) -> List[List[Tuple[int, np.float32]]]:
    """computeResizeAreaTab (resize.cpp): per destination index, the (source index, weight) list."""
    tab = []
    for d in range(dsize):
        fsx1 = d * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        row = []
        if sx1 - fsx1 > 1e-3:
            row.append((sx1 - 1, f32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            row.append((sx, f32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            row.append((sx2, f32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        tab.append(row)
    return tab


def _linear_area_tab(ssize: int, dsize: int, scale: float, inv_scale: float):
    """Coefficients of the INTER_LINEAR path in area mode (what INTER_AREA becomes when enlarging),
    11-bit fixed point."""
    ofs, coef = np.zeros(dsize, np.int64), np.zeros((dsize, 2), np.int64)
    for d in range(dsize):
        s = math.floor(d * scale)
        fx = f32((d + 1) - (s + 1) * inv_scale)
        fx = f32(0.0) if fx <= 0 else f32(fx - f32(math.floor(fx)))
        if s < 0:
            fx, s = f32(0.0), 0
        if s >= ssize - 1:
            fx, s = f32(0.0), ssize - 1
        ofs[d] = s
        c0, c1 = f32(1.0) - fx, fx
        coef[d, 0] = int(np.clip(np.rint(f32(c0 * f32(2048.0))), -32768, 32767))
        coef[d, 1] = int(np.clip(np.rint(f32(c1 * f32(2048.0))), -32768, 32767))
    return ofs, coef


def resize_mode(sw: int, sh: int, dw: int, dh: int) -> str:
    """Which of cv::resize(INTER_AREA)'s code paths a shape takes."""
    if (sw, sh) == (dw, dh):
        return "copy"
    inv_x, inv_y = dw / sw, dh / sh
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    ix, iy = int(np.rint(scale_x)), int(np.rint(scale_y))
    fast = abs(scale_x - ix) < np.finfo(np.float64).eps and abs(scale_y - iy) < np.finfo(np.float64).eps
    if scale_x >= 1 and scale_y >= 1:
        return "area_fast" if fast else "area"
    return "linear"


def resize_area_u8(img: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (dw, dh), interpolation=cv2.INTER_AREA) for 8-bit 3-channel images."""
    sh, sw = img.shape[:2]
    dw, dh = dsize
    mode = resize_mode(sw, sh, dw, dh)
    if mode == "copy":
        return img.copy()
    inv_x, inv_y = dw / sw, dh / sh
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    if mode == "area_fast":
        ix, iy = int(np.rint(scale_x)), int(np.rint(scale_y))
        blk = img[:dh * iy, :dw * ix].astype(np.int64).reshape(dh, iy, dw, ix, -1).sum(axis=(1, 3))
        if ix == 2 and iy == 2:
            return ((blk + 2) >> 2).astype(np.uint8)
        val = blk.astype(f32) * f32(1.0 / (ix * iy))
        return np.clip(np.rint(val), 0, 255).astype(np.uint8)
    if mode == "area":
        xtab, ytab = _area_tab(sw, dw, scale_x), _area_tab(sh, dh, scale_y)
        src = img.astype(f32)
        kx = max(len(r) for r in xtab)
        buf = np.zeros((sh, dw, img.shape[2]), f32)
        for k in range(kx):  # sequential float32 accumulation in tab order
            si = np.array([r[k][0] if k < len(r) else 0 for r in xtab])
            al = np.array([r[k][1] if k < len(r) else 0 for r in xtab], dtype=f32)
            on = np.array([k < len(r) for r in xtab])
            term = (src[:, si, :] * al[None, :, None]).astype(f32)
            buf = np.where(on[None, :, None], (buf + term).astype(f32), buf)
        ky = max(len(r) for r in ytab)
        acc = np.zeros((dh, dw, img.shape[2]), f32)
        for k in range(ky):
            si = np.array([r[k][0] if k < len(r) else 0 for r in ytab])
            be = np.array([r[k][1] if k < len(r) else 0 for r in ytab], dtype=f32)
            on = np.array([k < len(r) for r in ytab])
            term = (buf[si] * be[:, None, None]).astype(f32)
            acc = np.where(on[:, None, None], (acc + term).astype(f32), acc)
        return np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    # enlarging (in at least one direction): bilinear with area-mode coefficients, fixed point
    xo, xc = _linear_area_tab(sw, dw, scale_x, inv_x)
    yo, yc = _linear_area_tab(sh, dh, scale_y, inv_y)
    src = img.astype(np.int64)
    x1 = np.minimum(xo + 1, sw - 1)
    hrow = src[:, xo, :] * xc[None, :, 0, None] + src[:, x1, :] * xc[None, :, 1, None]  # [sh, dw, c]
    y1 = np.minimum(yo + 1, sh - 1)
    s0, s1 = hrow[yo], hrow[y1]
    b0, b1 = yc[:, 0, None, None], yc[:, 1, None, None]
    out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


_HSV_SHIFT = 12
_SDIV = np.array([0] + [int(np.rint((255 << _HSV_SHIFT) / (1.0 * i))) for i in range(1, 256)], dtype=np.int64)
_HDIV180 = np.array([0] + [int(np.rint((180 << _HSV_SHIFT) / (6.0 * i))) for i in range(1, 256)], dtype=np.int64)


def bgr2hsv_u8(img: np.ndarray) -> np.ndarray:
    """cvtColor(COLOR_BGR2HSV) on 8-bit data (H in [0,180)): integer arithmetic with division tables.
    Channel 0 is treated as blue whatever it really holds (the reference feeds RGB images)."""
    b, g, r = (img[..., i].astype(np.int64) for i in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    s = (diff * _SDIV[v] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV180[diff] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


def hsv2bgr_u8(img: np.ndarray) -> np.ndarray:
    """cvtColor(COLOR_HSV2BGR) on 8-bit data: float32 sector arithmetic, rounded back to 8 bits."""
    h = img[..., 0].astype(f32)
    s = (img[..., 1].astype(f32) * f32(1.0 / 255.0)).astype(f32)
    v = (img[..., 2].astype(f32) * f32(1.0 / 255.0)).astype(f32)
    hh = (h * f32(6.0 / 180.0)).astype(f32)
    hh = np.where(hh >= 6, (hh - f32(6.0)).astype(f32), hh)
    sector = np.floor(hh).astype(np.int64)
    frac = (hh - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    frac = np.where(bad, f32(0), frac)
    one = f32(1.0)
    t0 = v
    t1 = (v * (one - s)).astype(f32)
    t2 = (v * (one - (s * frac).astype(f32)).astype(f32)).astype(f32)
    t3 = (v * (one - (s * (one - frac).astype(f32)).astype(f32)).astype(f32)).astype(f32)
    tab = np.stack([t0, t1, t2, t3], axis=-1)
    sector_data = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    idx = sector_data[sector]  # [..., 3] -> indices for (b, g, r)
    bgr = np.take_along_axis(tab, idx, axis=-1)
    bgr = np.where((s == 0)[..., None], v[..., None], bgr)
    return np.clip(np.rint((bgr * f32(255.0)).astype(f32)), 0, 255).astype(np.uint8)


def color_jitter_u8(img: np.ndarray, h: float, s: float, a: float, b: float) -> np.ndarray:
    """sample_augmenter.py:281-293: scale H and S, affine V (float64, clipped to [0,255], truncated)."""
    hsv = bgr2hsv_u8(img)
    hue = np.clip(hsv[..., 0] * h, 0, 255)
    sat = np.clip(hsv[..., 1] * s, 0, 255)
    val = np.clip(hsv[..., 2] * a + b, 0, 255)
    return hsv2bgr_u8(np.stack([hue, sat, val], axis=-1).astype(np.uint8))


def to_tensor_normalize(img: np.ndarray) -> np.ndarray:
    """ToTensor + Normalize (data_loader/utils.py:285-291): float32, CHW."""
    x = (img.transpose(2, 0, 1).astype(f32) / f32(255.0)).astype(f32)
    mean = np.array(IMAGENET_MEAN, dtype=f32)[:, None, None]
    std = np.array(IMAGENET_STD, dtype=f32)[:, None, None]
    return ((x - mean).astype(f32) / std).astype(f32)


def render_view(image: np.ndarray, view: Dict, flags: Dict[str, bool], params: Dict, stages: bool = False):
    """Pixel side of transform_sample for parameters drawn by `sample_view`."""
    img = image
    out = {}
    if view["rot"] is not None:
        img = warp_affine_u8(img, view["rot"])
        out["rotated"] = img
    if view["crop"] is not None:
        x0, y0, cw, ch = view["crop"]
        img = img[y0:y0 + ch, x0:x0 + cw]
        out["cropped"] = img
    if flags.get("resize") and view["resized"]:
        img = resize_area_u8(img, tuple(params["resize_shape"]))
        out["resized"] = img
    if flags.get("color_jitter"):
        img = color_jitter_u8(img, view["h"], view["s"], view["a"], view["b"])
        out["jittered"] = img
    out["tensor"] = to_tensor_normalize(img)
    return out if stages else out["tensor"]


def prepare_hybrid2_sample(image: np.ndarray, joints25d: np.ndarray, flags: Dict[str, bool], params: Dict, rng) -> Dict:
    """data_set.py:357-384: two independently drawn views of one image + the parameter entries that
    are not None (keys suffixed _1 / _2)."""
    override = None if flags.get("crop") else [0, 0]
    out = {}
    for i in (1, 2):
        view = sample_view(joints25d, image.shape[:2], flags, params, rng, override)
        out[f"transformed_image{i}"] = render_view(image, view, flags, params)
        entries = {"angle": view["angle"], "jitter_x": view["jitter_x"], "jitter_y": view["jitter_y"], "h": view["h"],
                   "s": view["s"], "a": view["a"], "b": view["b"], "blur_flag": view["blur_flag"],
                   "crop_margin_scale": view["crop_margin_scale"]}
        out.update({f"{k}_{i}": v for k, v in entries.items() if v is not None})
    return out
