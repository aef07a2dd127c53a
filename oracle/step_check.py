"""Checker (test infrastructure, like everything under oracle/): run the HIP head + alignment +
NT-Xent of a `Hybrid2Model` on one batch, run the NumPy oracle on the SAME encoder output, and
report the differences `BASELINE.json:metric` names ("NT-Xent loss delta vs ref"; north_star:
<= 1e-4 fp32 on the loss and the per-pair similarities).

Used by tests/ (`-m gpu`), `__graft_entry__.smoke()` and bench.py's parity line -- never by the
product.  The encoder output `h` is taken from the model's own forward (hook), so the comparison
covers exactly the hand-written part of the step (reference chain: hybrid2_model.py:27-90,
utils.py:154-186) at whatever encoder / batch size the caller built.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import peclr_oracle as O


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def head_params(model) -> Dict[str, np.ndarray]:
    ph = model.projection_head
    return {k: _np(v).copy() for k, v in dict(w1=ph[0].weight, b1=ph[0].bias, gamma=ph[1].weight, beta=ph[1].bias,
                                              w2=ph[3].weight).items()}


def oracle_on(h: np.ndarray, w: Dict[str, np.ndarray], batch: Dict[str, torch.Tensor], augmentation,
              relu_ties=None) -> dict:
    """The reference's head + alignment + loss (+ closed-form backward) on encoder output `h` [2N, Din]."""
    n = h.shape[0] // 2
    crop, rotate = "crop" in augmentation, "rotate" in augmentation
    hw = tuple(int(v) for v in batch["transformed_image1"].shape[-2:])
    kw = {}
    if crop:
        kw["jitter_x"] = torch.cat([batch["jitter_x_1"], batch["jitter_x_2"]]).cpu().numpy()
        kw["jitter_y"] = torch.cat([batch["jitter_y_1"], batch["jitter_y_2"]]).cpu().numpy()
    if rotate:
        kw["angle"] = torch.cat([batch["angle_1"], batch["angle_2"]]).cpu().numpy()
    # float64 throughout: the comparison then measures the HIP path's own fp32 round-off, not the sum of two
    # fp32 evaluations with different summation orders (BatchNorm1d's backward amplifies both)
    f64 = {k: v.astype(np.float64) for k, v in w.items()}
    return O.head_loss_fwd_bwd(h.astype(np.float64), f64["w1"], f64["b1"], f64["gamma"], f64["beta"], f64["w2"], n,
                               crop=crop, rotate=rotate, image_hw=hw, relu_ties=relu_ties, **kw)


def step_deltas(model, batch: Dict[str, torch.Tensor], autocast=None, backward: bool = False) -> dict:
    """HIP path vs oracle on `batch` (this rank's rows only: local negatives, whatever the world size).

    Returns loss / similarity / embedding deltas (+ `dh` deltas and the oracle's gradients with
    backward=True).  The forward runs in the model's current mode; with backward=False it runs under
    no_grad and leaves parameters and gradients untouched (BatchNorm running statistics move once, as in
    any training-mode forward)."""
    from peclr_amd import _capi

    feats = {}
    hook = model.encoder.register_forward_hook(lambda m, i, o: feats.__setitem__("h", o))
    w = head_params(model)
    ctx = autocast if autocast is not None else torch.autocast("cuda", enabled=False)
    # the hidden activation of the HIP head (post-ReLU), to settle rectifier ties (see oracle.projection_head_fwd)
    real_bn_relu = _capi.bn_relu_fwd

    def spy_bn_relu(*a, **k):
        out = real_bn_relu(*a, **k)
        feats["a"] = out[1]
        return out

    _capi.bn_relu_fwd = spy_bn_relu
    try:
        with torch.set_grad_enabled(backward), ctx:
            z, row_stats, n = model._project(batch)
        h_t = feats["h"]
        if backward:
            h_t.retain_grad()
    finally:
        hook.remove()
        _capi.bn_relu_fwd = real_bn_relu
    ties = (_np(feats["a"]) > 0) if "a" in feats else None
    zc = z.detach().contiguous()
    m = zc.shape[0]
    out17, _lse, sim = _capi.ntxent_fwd(zc, 0, zc, n, 1.0 / 0.5, 1.0 / m,
                                        row_stats if row_stats.numel() else None, n, want_sim=True)
    ref = oracle_on(_np(h_t), w, batch, model.config.augmentation, relu_ties=ties)
    res = {"loss_hip": float(out17[16]), "loss_oracle": float(ref["loss"]),
           "loss_delta_vs_oracle": abs(float(out17[16]) - float(ref["loss"])),
           "sim_max_abs_delta": float(np.abs(_np(sim) - ref["sim"]).max()),
           "z_max_abs_delta": float(np.abs(_np(z) - ref["z"]).max()),
           "rows": int(m), "encoder_dim": int(h_t.shape[1]), "relu_tie_count": ref["relu_tie_count"]}
    if row_stats.numel():
        res["stats_max_abs_delta"] = float(np.abs(_np(out17[:16]) - np.asarray(ref["stats"], np.float64)).max())
    if backward:
        from peclr_amd import ops

        loss, _, _ = ops.ntxent(z, n, 0.5, None, None)   # NOTE: global negatives when a process group is live
        loss.backward()
        dh = _np(h_t.grad)
        scale = max(1e-30, float(np.abs(ref["dh"]).max()))
        res.update(dh_max_abs_delta=float(np.abs(dh - ref["dh"]).max()), dh_scale=scale,
                   dh_rel=float(np.abs(dh - ref["dh"]).max()) / scale, oracle=ref)
    return res
