"""CPU oracle for the PeCLR pretraining hot path -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference's algorithm for the path that
`BASELINE.json:north_star` names (projection head -> projection stats ->
normalise -> un-translate -> un-rotate -> normalise -> NT-Xent, forward and
closed-form backward).  It is the *checker*: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product (`peclr_amd/`) never imports it and has no CPU fallback.

Pinning: the reference has no tests or known-answer vectors of its own
(SURVEY.md section 4), so this oracle is pinned against golden vectors captured
in the build container from the reference's own functions imported under
third-party stubs (`tests/golden/make_golden.py`, fixtures `tests/golden/*.npz`).
`tests/test_oracle_golden.py` holds that check.  Two pieces have no importable
reference: torchvision 0.8's ResNet -- the in-tree restatement
(`peclr_amd/resnet.py`) is instead pinned against an INDEPENDENT implementation of the
same published architecture, `transformers.ResNetModel` (equal to 1e-10 in float64 with
copied weights, eval and train mode: tests/test_host_logic.py) -- and pl_bolts'
LARSWrapper / LinearWarmupCosineAnnealingLR, restated from their published behaviour in
`lars_adam_step` / `warmup_cosine_lr` and "parity unpinned" (nothing implementing them is
installed); the numbers the reference feeds them are pinned by golden G7.
`bench.py` also uses `oracle/step_check.py` for its parity line (checker, untimed).

All citations are `path:line` relative to /root/reference.

Every function computes in the dtype of its floating inputs: float32 inputs
reproduce the reference's fp32 arithmetic (up to summation order), float64 inputs
give a high-precision "truth" used for error budgeting.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np

F_EPS = 1e-12  # torch.nn.functional.normalize default eps (hybrid2_model.py:48-49,83-84)
BN_EPS = 1e-5  # nn.BatchNorm1d default eps (simclr_model.py:27)
BN_MOMENTUM = 0.1  # nn.BatchNorm1d default momentum


# --------------------------------------------------------------------------- #
# K1 / K2: projection head  (simclr_model.py:20-35)
# --------------------------------------------------------------------------- #
def linear_fwd(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    """nn.Linear: y = x @ w.T + b   (simclr_model.py:22-26, 29-33)."""
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


def bn1d_train_fwd(a: np.ndarray, gamma: np.ndarray, beta: np.ndarray, eps: float = BN_EPS):
    """nn.BatchNorm1d in training mode (simclr_model.py:27).

    Batch statistics over ALL rows (both views together, hybrid2_model.py:30-38);
    biased variance for the normalisation.  Returns y, (mean, var_biased, invstd, xhat).
    """
    mean = a.mean(axis=0)
    var = ((a - mean) ** 2).mean(axis=0)
    invstd = 1.0 / np.sqrt(var + np.asarray(eps, a.dtype))
    xhat = (a - mean) * invstd
    y = xhat * gamma + beta
    return y, (mean, var, invstd, xhat)


def bn1d_running_update(running_mean, running_var, mean, var_biased, m_rows: int,
                        momentum: float = BN_MOMENTUM):
    """Running-stat update of nn.BatchNorm1d: unbiased variance, momentum 0.1."""
    unbiased = var_biased * (m_rows / max(m_rows - 1, 1))
    mom = np.asarray(momentum, running_mean.dtype)
    new_mean = (1 - mom) * running_mean + mom * mean
    new_var = (1 - mom) * running_var + mom * unbiased
    return new_mean, new_var


RELU_TIE = 1e-5  # |pre-activation| below which another fp32 evaluation may rectify the other way


def projection_head_fwd(h, w1, b1, gamma, beta, w2, eps: float = BN_EPS, relu_ties=None):
    """Linear(Din,H,bias) -> BatchNorm1d(H) -> ReLU -> Linear(H,D,no bias).

    simclr_model.py:20-35.  Returns p [M,D] and a cache for the backward.
    relu_ties (checker use only): bool [M,H], the rectifier decisions of the implementation under test.
    They are adopted ONLY where this evaluation's pre-activation is within RELU_TIE of zero -- there the
    sign is decided by fp32 round-off, both choices are "the reference's", and the two backward passes
    are only comparable if they agree on it; everywhere else the oracle's own decision stands, so a wrong
    mask in the implementation still shows up.
    """
    a_pre = linear_fwd(h, w1, b1)
    y, (mean, var, invstd, xhat) = bn1d_train_fwd(a_pre, gamma, beta, eps)
    on = y > 0
    if relu_ties is not None:
        on = np.where(np.abs(y) < RELU_TIE, np.asarray(relu_ties, bool), on)
    a = np.where(on, y, 0)
    p = linear_fwd(a, w2)
    cache = dict(h=h, w1=w1, w2=w2, gamma=gamma, a_pre=a_pre, mean=mean, var=var,
                 invstd=invstd, xhat=xhat, y=y, a=a, on=on)
    return p, cache


def projection_head_bwd(dp, cache):
    """Closed-form backward of `projection_head_fwd` (autograd of simclr_model.py:20-35).

    Returns dict(dh, dw1, db1, dgamma, dbeta, dw2).
    """
    h, w1, w2 = cache["h"], cache["w1"], cache["w2"]
    gamma, invstd, xhat, a, on = (cache[k] for k in ("gamma", "invstd", "xhat", "a", "on"))
    m = h.shape[0]
    dw2 = dp.T @ a
    da = dp @ w2
    dy = da * on
    dbeta = dy.sum(axis=0)
    dgamma = (dy * xhat).sum(axis=0)
    da_pre = (gamma * invstd / m) * (m * dy - dbeta - xhat * dgamma)
    dw1 = da_pre.T @ h
    db1 = da_pre.sum(axis=0)
    dh = da_pre @ w1
    return dict(dh=dh, dw1=dw1, db1=db1, dgamma=dgamma, dbeta=dbeta, dw2=dw2)


# --------------------------------------------------------------------------- #
# K3: projection statistics  (hybrid2_model.py:92-106, called :40-45)
# --------------------------------------------------------------------------- #
STAT_NAMES = ("x_mean", "x_median", "x_min", "x_max", "y_mean", "y_median", "y_min", "y_max")


def projection_stats(p_view: np.ndarray, name: str) -> Dict[str, np.ndarray]:
    """p_view [N,P,2] (one view).  torch.median = LOWER median: sorted[(P-1)//2]."""
    n, npts, _ = p_view.shape
    mean = p_view.mean(axis=1)
    med = np.sort(p_view, axis=1)[:, (npts - 1) // 2, :]
    mn = p_view.min(axis=1)
    mx = p_view.max(axis=1)
    out = {}
    for c, axis_name in enumerate(("x", "y")):
        out[f"{name}{axis_name}_mean"] = mean[:, c].mean()
        out[f"{name}{axis_name}_median"] = med[:, c].mean()
        out[f"{name}{axis_name}_min"] = mn[:, c].mean()
        out[f"{name}{axis_name}_max"] = mx[:, c].mean()
    return out


def stats_vector(p: np.ndarray, n_pairs: int) -> np.ndarray:
    """The 16 scalars in the order the HIP kernel emits them:
    [proj1: x_mean,x_median,x_min,x_max,y_mean,y_median,y_min,y_max, proj2: same]."""
    m = p.shape[0]
    pv = p.reshape(m, -1, 2)
    out = []
    for name, sl in (("proj1", slice(0, n_pairs)), ("proj2", slice(n_pairs, m))):
        d = projection_stats(pv[sl], name)
        out += [d[f"{name}{s}"] for s in STAT_NAMES]
    return np.asarray(out, p.dtype)


def stat_keys() -> Tuple[str, ...]:
    return tuple(f"{n}{s}" for n in ("proj1", "proj2") for s in STAT_NAMES)


# --------------------------------------------------------------------------- #
# K4 / K7: F.normalize  (hybrid2_model.py:48-49, 83-84; simclr_model.py:44-47)
# --------------------------------------------------------------------------- #
def l2_normalize_fwd(x: np.ndarray, eps: float = F_EPS):
    """x / max(||x||_2, eps) per row.  Returns (y, clamped_norm[M])."""
    n = np.sqrt((x * x).sum(axis=1))
    nc = np.maximum(n, np.asarray(eps, x.dtype))
    return x / nc[:, None], nc


def l2_normalize_bwd(dy: np.ndarray, y: np.ndarray, nc: np.ndarray, eps: float = F_EPS):
    """dx = (dy - y (y.dy)) / ||x||   (dx = dy/eps where the norm was clamped)."""
    dot = (dy * y).sum(axis=1, keepdims=True)
    clamped = (nc <= np.asarray(eps, y.dtype))[:, None]
    return np.where(clamped, dy / nc[:, None], (dy - y * dot) / nc[:, None])


# --------------------------------------------------------------------------- #
# K5: translate_encodings  (utils.py:325-346, called hybrid2_model.py:58-74)
# --------------------------------------------------------------------------- #
def jitter_to_translation(jitter: np.ndarray, image_extent: int, dtype=np.float32):
    """-(int64 jitter / float(extent)) as the model forms it (hybrid2_model.py:59-74).

    torch divides an int64 tensor by a Python float in the default dtype (fp32)."""
    return -(jitter.astype(dtype) / np.asarray(float(image_extent), dtype))


def translate_fwd(q: np.ndarray, tx: np.ndarray, ty: np.ndarray) -> np.ndarray:
    """q [M,P,2]; per-sample range over the P points is a CONSTANT (detached)."""
    rng = q.max(axis=1) - q.min(axis=1)  # [M,2]
    out = q.copy()
    out[..., 0] += (tx * rng[:, 0])[:, None]
    out[..., 1] += (ty * rng[:, 1])[:, None]
    return out


# --------------------------------------------------------------------------- #
# K6: rotate_encoding + get_rotation_2D_matrix  (utils.py:271-321)
# --------------------------------------------------------------------------- #
def rotation_matrix(angle_deg: np.ndarray, cx: np.ndarray, cy: np.ndarray) -> np.ndarray:
    """OpenCV-style 3x2 matrix, built in float64 then stored float32 (utils.py:287-298).

    `angle_deg` is what `rotate_encoding` receives, i.e. ALREADY negated by the
    model (hybrid2_model.py:80)."""
    theta = angle_deg.astype(np.float64) * np.pi / 180
    alpha = np.cos(theta)
    beta = np.sin(theta)
    cx64 = cx.astype(np.float64)
    cy64 = cy.astype(np.float64)
    r = np.zeros((len(theta), 3, 2), np.float32)
    r[:, :, 0] = np.stack([alpha, beta, (1 - alpha) * cx64 - beta * cy64], axis=1)
    r[:, :, 1] = np.stack([-beta, alpha, (1 - alpha) * cy64 + beta * cx64], axis=1)
    return r


def rotate_fwd(q: np.ndarray, angle_deg: np.ndarray):
    """q [M,P,2] -> [x,y,1] @ R about the per-sample centroid (detached).

    Returns (out, R)."""
    c = q.mean(axis=1)
    r = rotation_matrix(angle_deg, c[:, 0], c[:, 1]).astype(q.dtype)
    ones = np.ones(q.shape[:2] + (1,), q.dtype)
    out = np.einsum("mpk,mkc->mpc", np.concatenate([q, ones], axis=2), r)
    return out, r


def rotate_bwd(dout: np.ndarray, r: np.ndarray) -> np.ndarray:
    """Centroid is a constant, so only the 2x2 linear part back-propagates."""
    return np.einsum("mpc,mkc->mpk", dout, r[:, :2, :])


# --------------------------------------------------------------------------- #
# K3-K7 composite: what Hybrid2Model.get_transformed_projections does after
# the head (hybrid2_model.py:40-85)
# --------------------------------------------------------------------------- #
def align_fwd(p: np.ndarray, n_pairs: int, *, crop: bool, rotate: bool,
              jitter_x: Optional[np.ndarray] = None, jitter_y: Optional[np.ndarray] = None,
              angle: Optional[np.ndarray] = None, image_hw: Tuple[int, int] = (224, 224),
              double_norm: bool = True):
    """p [M,D] raw projections (rows: view-1 samples then view-2 samples).

    jitter_x/jitter_y: int64 [M] (cat of the two views), angle: float64 [M] as the
    batch holds them (NOT negated).  `double_norm=False` is SimCLR.contrastive_step
    (simclr_model.py:37-49: a single F.normalize, no alignment).
    Returns z [M,D], stats[16], cache.
    """
    m, d = p.shape
    stats = stats_vector(p, n_pairs)
    q, n1 = l2_normalize_fwd(p)
    cache = dict(q0=q, n1=n1, crop=crop, rotate=rotate, double_norm=double_norm, r=None)
    if not double_norm:
        cache.update(z=q, n2=None)
        return q, stats, cache
    u = q.reshape(m, d // 2, 2)
    if crop:
        tx = jitter_to_translation(jitter_x, image_hw[0], p.dtype)  # x / shape[0] (quirk kept)
        ty = jitter_to_translation(jitter_y, image_hw[1], p.dtype)
        u = translate_fwd(u, tx, ty)
    if rotate:
        u, r = rotate_fwd(u, -angle.astype(np.float64))
        cache["r"] = r
    z, n2 = l2_normalize_fwd(u.reshape(m, d))
    cache.update(z=z, n2=n2)
    return z, stats, cache


def align_bwd(dz: np.ndarray, cache) -> np.ndarray:
    """dL/dp given dL/dz; range and centroid are constants (utils.py:312,338-339)."""
    m, d = dz.shape
    if not cache["double_norm"]:
        return l2_normalize_bwd(dz, cache["q0"], cache["n1"])
    du = l2_normalize_bwd(dz, cache["z"], cache["n2"])
    if cache["rotate"]:
        du = rotate_bwd(du.reshape(m, d // 2, 2), cache["r"].astype(dz.dtype)).reshape(m, d)
    # translate: identity
    return l2_normalize_bwd(du, cache["q0"], cache["n1"])


# --------------------------------------------------------------------------- #
# K8: vanila_contrastive_loss  (utils.py:154-186)
# --------------------------------------------------------------------------- #
def pair_index(m_rows: int, n_half: int) -> np.ndarray:
    """Positive partner of every row for the layout [.., view1 x n_half, view2 x n_half, ..]
    repeated m_rows/(2 n_half) times (one repeat per rank; a single repeat is the
    reference's layout cat(z1, z2), utils.py:171)."""
    idx = np.arange(m_rows)
    view = (idx // n_half) & 1
    return np.where(view == 1, idx - n_half, idx + n_half)


def ntxent_fwd(z: np.ndarray, n_half: int, temperature: float = 0.5):
    """z [M,D] unit rows.  Returns (loss, S [M,M], lse [M] = log sum_{j!=i} exp(S_ij/tau), pos [M]).

    Follows utils.py:171-186 literally: no max subtraction, denominator includes
    the positive and excludes only the diagonal, positive logit is an elementwise
    dot (utils.py:183), loss = -mean log(pos/neg).
    """
    m = z.shape[0]
    inv_tau = np.asarray(1.0 / temperature, z.dtype)
    s = z @ z.T
    e = np.exp(s * inv_tau)
    e_off = e.copy()
    np.fill_diagonal(e_off, 0)  # masked_select(~eye) drops exactly the diagonal (utils.py:179-180)
    neg = e_off.sum(axis=1)
    pidx = pair_index(m, n_half)
    pos_logit = (z * z[pidx]).sum(axis=1) * inv_tau
    lse = np.log(neg)
    loss = (lse - pos_logit).mean()
    return loss, s, lse, pos_logit


def ntxent_bwd(z: np.ndarray, lse: np.ndarray, n_half: int, temperature: float = 0.5,
               dloss: float = 1.0, rows: Optional[slice] = None, m_total: Optional[int] = None):
    """dL/dz = (G + G^T) z with G = (P - Y) / (M tau)  (SURVEY.md section 8a).

    Written in the symmetric per-row form the HIP kernel uses:
      dz_i = dloss/(M tau) * sum_{j != i} [ e_ij (1/neg_i + 1/neg_j) - 2 Y_ij ] z_j
    `rows` restricts the output to a row block (multi-GPU: local rows vs all columns).
    """
    m = z.shape[0] if m_total is None else m_total
    inv_tau = np.asarray(1.0 / temperature, z.dtype)
    rs = rows if rows is not None else slice(0, z.shape[0])
    zi = z[rs]
    gi = np.arange(z.shape[0])[rs]
    s = zi @ z.T
    e = np.exp(s * inv_tau)
    rinv = np.exp(-lse)
    w = e * (rinv[gi][:, None] + rinv[None, :])
    pidx = pair_index(z.shape[0], n_half)
    w[np.arange(len(gi)), gi] = 0
    w[np.arange(len(gi)), pidx[gi]] -= 2
    scale = np.asarray(dloss, z.dtype) * inv_tau / np.asarray(m, z.dtype)
    return (w @ z) * scale


# --------------------------------------------------------------------------- #
# Whole step after the encoder: head -> align -> NT-Xent, forward + backward
# (Hybrid2Model.contrastive_step hybrid2_model.py:87-90; SimCLR simclr_model.py:37-49)
# --------------------------------------------------------------------------- #
def head_loss_fwd_bwd(h, w1, b1, gamma, beta, w2, n_pairs: int, *, crop=False, rotate=False,
                      jitter_x=None, jitter_y=None, angle=None, image_hw=(224, 224),
                      temperature: float = 0.5, double_norm: bool = True, relu_ties=None):
    p, hc = projection_head_fwd(h, w1, b1, gamma, beta, w2, relu_ties=relu_ties)
    z, stats, ac = align_fwd(p, n_pairs, crop=crop, rotate=rotate, jitter_x=jitter_x,
                             jitter_y=jitter_y, angle=angle, image_hw=image_hw,
                             double_norm=double_norm)
    loss, s, lse, pos = ntxent_fwd(z, n_pairs, temperature)
    dz = ntxent_bwd(z, lse, n_pairs, temperature)
    dp = align_bwd(dz, ac)
    grads = projection_head_bwd(dp, hc)
    return dict(loss=loss, sim=s, lse=lse, z=z, p=p, stats=stats, dz=dz, dp=dp, **grads,
                bn_mean=hc["mean"], bn_var=hc["var"], a_pre=hc["a_pre"],
                relu_tie_count=int((np.abs(hc["y"]) < RELU_TIE).sum()))


# --------------------------------------------------------------------------- #
# Optimiser plumbing  (base_model.py:30-104).  pl_bolts 0.2.2 is not importable
# here and is not vendored in the reference: PARITY UNPINNED, restated from the
# published behaviour of LARSWrapper / LinearWarmupCosineAnnealingLR.
# --------------------------------------------------------------------------- #
def exclude_from_wt_decay(names, skip_list=("bias", "bn")):
    """base_model.py:30-51: substring match on the parameter NAME."""
    decay, no_decay = [], []
    for n in names:
        (no_decay if any(s in n for s in skip_list) else decay).append(n)
    return decay, no_decay


def effective_lr(lr: float, batch_size: int, num_of_mini_batch: int) -> float:
    """base_model.py:62-66."""
    return lr * math.sqrt(batch_size * num_of_mini_batch)


def schedule_lengths(warmup_epochs, max_epochs, train_iters_per_epoch, num_of_mini_batch):
    """base_model.py:67-88: 'epochs' are optimiser steps."""
    return (warmup_epochs * train_iters_per_epoch // num_of_mini_batch,
            max_epochs * train_iters_per_epoch // num_of_mini_batch)


def warmup_cosine_lr(step: int, base_lr: float, warmup: int, max_steps: int,
                     warmup_start_lr: float = 0.0, eta_min: float = 0.0) -> float:
    """Closed form of pl_bolts LinearWarmupCosineAnnealingLR (`_get_closed_form_lr`)."""
    if step < warmup:
        return warmup_start_lr + step * (base_lr - warmup_start_lr) / max(warmup - 1, 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (
        1 + math.cos(math.pi * (step - warmup) / max(max_steps - warmup, 1)))


def lars_adam_step(p, g, m, v, step: int, lr: float, weight_decay: float,
                   eta: float = 0.02, lars_eps: float = 1e-8, clip: bool = True,
                   betas=(0.9, 0.999), adam_eps: float = 1e-8):
    """One LARSWrapper(Adam) update of a single parameter tensor.

    LARSWrapper.step (pl_bolts 0.2.2): for every param with a grad, if both norms are
    non-zero: trust = eta*|p| / (|g| + wd*|p| + eps); if clip: trust = min(trust/lr, 1);
    g += wd*p; g *= trust; then the wrapped Adam steps with weight_decay forced to 0.
    Returns (p_new, m_new, v_new, g_used).  `step` is the 1-based Adam step count.
    """
    p_norm = np.sqrt((p.astype(np.float64) ** 2).sum())
    g_norm = np.sqrt((g.astype(np.float64) ** 2).sum())
    g = g.copy()
    if p_norm != 0 and g_norm != 0:
        trust = eta * p_norm / (g_norm + p_norm * weight_decay + lars_eps)
        if clip:
            trust = min(trust / lr, 1.0) if lr > 0 else 1.0
        g = ((g + weight_decay * p) * trust).astype(p.dtype)
    b1, b2 = betas
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + adam_eps
    p_new = p - (lr / bc1) * (m / denom)
    return p_new.astype(p.dtype), m.astype(p.dtype), v.astype(p.dtype), g
