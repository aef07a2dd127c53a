"""Pin the CPU oracle (oracle/peclr_oracle.py) against golden vectors captured from the
reference's own functions (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import peclr_oracle as O
from tests.conftest import GOLDEN, load_golden

TOL = 2e-6  # fp32 restatement vs fp32 reference: summation-order noise only


def close(a, b, tol=TOL):
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


@pytest.mark.parametrize("name", ["g1_ntxent_N2.npz", "g1_ntxent_N8.npz", "g1_ntxent_N32.npz",
                                  "g1_ntxent_N8_tau01.npz"])
def test_ntxent_matches_reference(name):
    g = load_golden(name)
    z = np.concatenate([g["z1"], g["z2"]])
    n = len(g["z1"])
    tau = float(g["temperature"])
    loss, s, lse, pos = O.ntxent_fwd(z, n, tau)
    close(loss, g["loss"])
    close(s, g["sim"])
    dz = O.ntxent_bwd(z, lse, n, tau)
    close(dz[:n], g["dz1"], 5e-6)
    close(dz[n:], g["dz2"], 5e-6)
    # float64 truth agrees too (error budget of the fp32 reference itself)
    loss64, *_ = O.ntxent_fwd(z.astype(np.float64), n, tau)
    assert abs(loss64 - float(g["loss"])) < 2e-6


def test_ntxent_row_block_backward_equals_full():
    g = load_golden("g1_ntxent_N8.npz")
    z = np.concatenate([g["z1"], g["z2"]]).astype(np.float64)
    _, _, lse, _ = O.ntxent_fwd(z, 8)
    full = O.ntxent_bwd(z, lse, 8)
    blk = O.ntxent_bwd(z, lse, 8, rows=slice(4, 12))
    close(blk, full[4:12], 1e-12)


def test_rotate_matches_reference():
    g = load_golden("g2_rotate.npz")
    c = g["q"].mean(axis=1)
    r = O.rotation_matrix(g["angle"], c[:, 0], c[:, 1])
    close(r, g["rot_mat"], 1e-7)
    out, r2 = O.rotate_fwd(g["q"], g["angle"])
    close(out, g["out"])
    close(O.rotate_bwd(g["dout"], r2), g["dq"])


@pytest.mark.parametrize("size", [224, 448])
def test_translate_matches_reference(size):
    g = load_golden(f"g3_translate_{size}.npz")
    tx = O.jitter_to_translation(g["jitter_x"], size)
    ty = O.jitter_to_translation(g["jitter_y"], size)
    np.testing.assert_array_equal(tx, g["tx"])  # bit-exact fp32 division
    np.testing.assert_array_equal(ty, g["ty"])
    close(O.translate_fwd(g["q"], tx, ty), g["out"], 1e-7)


def _run_oracle_step(g, crop, rotate, double_norm=True):
    n = int(g["n_pairs"])
    kw = {}
    if crop:
        kw.update(jitter_x=np.concatenate([g["batch_jitter_x_1"], g["batch_jitter_x_2"]]),
                  jitter_y=np.concatenate([g["batch_jitter_y_1"], g["batch_jitter_y_2"]]))
    if rotate:
        kw.update(angle=np.concatenate([g["batch_angle_1"], g["batch_angle_2"]]))
    hw = tuple(int(v) for v in g["image_hw"]) if "image_hw" in g else (1, 1)
    return O.head_loss_fwd_bwd(g["h"], g["in_w1"], g["in_b1"], g["in_gamma"], g["in_beta"],
                               g["in_w2"], n, crop=crop, rotate=rotate, image_hw=hw,
                               double_norm=double_norm, **kw)


@pytest.mark.parametrize("tag,crop,rotate", [("none", False, False), ("crop", True, False),
                                             ("rotate", False, True),
                                             ("crop_rotate", True, True), ("wide", True, True)])
def test_hybrid2_step_matches_reference(tag, crop, rotate):
    g = load_golden(f"g4_hybrid2_{tag}.npz")
    r = _run_oracle_step(g, crop, rotate)
    close(r["loss"], g["loss"], 3e-6)
    keys = [str(k) for k in g["out_keys"]]
    assert keys == list(O.stat_keys()) + ["loss"]  # the 17 keys, in the reference's order
    for k, v in zip(O.stat_keys(), r["stats"]):
        close(v, g[f"out_{k}"], 3e-6)
    gtol = 2e-5  # grads pass through BN-train backward: cancellation amplifies fp32 noise
    for k in ("dh", "dw1", "dgamma", "dbeta", "dw2"):
        scale = max(1.0, float(np.abs(g[k]).max()))
        np.testing.assert_allclose(r[k], g[k], rtol=0, atol=gtol * scale, err_msg=k)
    # db1 is analytically zero (bias feeds a batch norm); both sides hold rounding noise
    assert np.abs(r["db1"]).max() < 1e-6 and np.abs(g["db1"]).max() < 1e-6
    m = g["h"].shape[0]
    rm, rv = O.bn1d_running_update(g["running_mean0"], g["running_var0"], r["bn_mean"],
                                   r["bn_var"], m)
    close(rm, g["running_mean1"], 1e-6)
    close(rv, g["running_var1"], 1e-6)
    assert int(g["num_batches_tracked1"]) == 1


def test_hybrid2_validation_step_surface():
    g = load_golden("g4_hybrid2_val.npz")
    assert [str(k) for k in g["out_keys"]] == ["loss"]
    # quirk kept: validation also overwrites train_metrics with the 16 stats
    assert [str(k) for k in g["train_metric_keys"]] == list(O.stat_keys())
    r = _run_oracle_step(g, True, True)
    close(r["loss"], g["loss"], 3e-6)


def test_simclr_step_matches_reference():
    g = load_golden("g5_simclr.npz")
    r = _run_oracle_step(g, False, False, double_norm=False)
    close(r["loss"], g["loss"], 3e-6)
    for k in ("dh", "dw1", "dgamma", "dbeta", "dw2"):
        scale = max(1.0, float(np.abs(g[k]).max()))
        np.testing.assert_allclose(r[k], g[k], rtol=0, atol=2e-5 * scale, err_msg=k)


def test_projection_head_matches_reference():
    g = load_golden("g6_head.npz")
    p, c = O.projection_head_fwd(g["h"], g["in_w1"], g["in_b1"], g["in_gamma"], g["in_beta"],
                                 g["in_w2"])
    close(p, g["p"], 5e-6)
    gr = O.projection_head_bwd(g["dp"], c)
    for k in ("dh", "dw1", "dgamma", "dbeta", "dw2"):
        scale = max(1.0, float(np.abs(g[k]).max()))
        np.testing.assert_allclose(gr[k], g[k], rtol=0, atol=1e-5 * scale, err_msg=k)
    m = g["h"].shape[0]
    rm, rv = O.bn1d_running_update(np.zeros_like(c["mean"]), np.ones_like(c["var"]), c["mean"],
                                   c["var"], m)
    close(rm, g["running_mean1"], 1e-6)
    close(rv, g["running_var1"], 1e-6)
    p2, c2 = O.projection_head_fwd(g["h2"], g["in_w1"], g["in_b1"], g["in_gamma"], g["in_beta"],
                                   g["in_w2"])
    close(p2, g["p2"], 5e-6)
    rm2, rv2 = O.bn1d_running_update(rm, rv, c2["mean"], c2["var"], m)
    close(rm2, g["running_mean2"], 1e-6)
    close(rv2, g["running_var2"], 1e-6)
    assert int(g["num_batches_tracked2"]) == 2


def test_optimizer_plumbing_matches_reference():
    with open(os.path.join(GOLDEN, "g7_optim.json")) as f:
        g = json.load(f)
    names = g["membership"]["decay"] + g["membership"]["no_decay"]
    decay, no_decay = O.exclude_from_wt_decay(sorted(names, key=names.index))
    assert sorted(decay) == sorted(g["membership"]["decay"])
    assert sorted(no_decay) == sorted(g["membership"]["no_decay"])
    for c in g["cases"]:
        assert O.effective_lr(1e-4, c["batch_size"], c["accum"]) == pytest.approx(c["lr"][0], rel=1e-12)
        iters = c["num_samples"] // (c["world_size"] * c["batch_size"])
        assert iters == c["train_iters_per_epoch"]
        max_ep = c["lr_max_epochs"] if c["lr_max_epochs"] is not None else c["trainer_max_epochs"]
        wu, mx = O.schedule_lengths(10, max_ep, iters, c["accum"])
        assert (wu, mx) == (c["warmup_epochs"], c["max_epochs"])


def test_projection_stats_lower_median():
    rng = np.random.default_rng(0)
    p = rng.standard_normal((4, 64, 2)).astype(np.float32)
    d = O.projection_stats(p, "proj1")
    med = np.sort(p[:, :, 0], axis=1)[:, 31].mean()
    assert d["proj1x_median"] == pytest.approx(med)
