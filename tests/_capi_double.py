"""TEST DOUBLE for peclr_amd._capi: every kernel wrapper re-implemented on CPU tensors with the
oracle, so the HOST logic above the C ABI (autograd wiring, module surface, data-parallel
plumbing, trainer) can be exercised without a GPU.  Installed only through pytest's monkeypatch;
the product has no switch that routes here.
"""
import numpy as np
import torch

from oracle import peclr_oracle as O


def _np(t):
    return t.detach().cpu().numpy()


def _t(a, like=None):
    return torch.from_numpy(np.ascontiguousarray(a))


def pick_split_k(m, n, k):
    return 2 if k >= 8 else 1  # always exercise the slab path


def gemm(layout, a, b, bias=None, split_k=1, tag=None):
    if layout == 0:
        c = a @ b.t()
    elif layout == 1:
        c = a @ b
    else:
        c = a.t() @ b
    if split_k == 1:
        return c + bias if bias is not None else c
    slabs = torch.zeros((split_k,) + tuple(c.shape))
    slabs[0], slabs[-1] = 0.25 * c, 0.75 * c
    return slabs


def slab_reduce(slabs, bias=None):
    out = slabs.sum(0)
    return out + bias if bias is not None else out


def bn_relu_fwd(a_slabs, bias, gamma, beta, eps, momentum, training, running_mean, running_var, nbt):
    a_pre = _np(a_slabs.sum(0) + bias)
    m = a_pre.shape[0]
    if training:
        y, (mean, var, invstd, _) = O.bn1d_train_fwd(a_pre, _np(gamma), _np(beta), eps)
        if running_mean is not None:
            rm, rv = O.bn1d_running_update(_np(running_mean), _np(running_var), mean, var, m, momentum)
            running_mean.copy_(_t(rm))
            running_var.copy_(_t(rv))
        if nbt is not None:
            nbt += 1
    else:
        mean, invstd = _np(running_mean), 1.0 / np.sqrt(_np(running_var) + np.float32(eps))
        y = (a_pre - mean) * invstd * _np(gamma) + _np(beta)
    save = torch.stack([_t(mean.astype(np.float32)), _t(invstd.astype(np.float32))])
    return _t(a_pre), _t(np.maximum(y, 0).astype(np.float32)), save


def bn_relu_bwd(d_a_out, a_pre, save, gamma, beta, training=True):
    x, mean, invstd = _np(a_pre), _np(save[0]), _np(save[1])
    g, b, da = _np(gamma), _np(beta), _np(d_a_out)
    m = x.shape[0]
    xhat = (x - mean) * invstd
    dy = da * ((xhat * g + b) > 0)
    dbeta, dgamma = dy.sum(0), (dy * xhat).sum(0)
    dx = (g * invstd / m) * (m * dy - dbeta - xhat * dgamma) if training else g * invstd * dy
    return _t(dx.astype(np.float32)), _t(dgamma), _t(dbeta), _t(dx.sum(0))


_ALIGN_CACHE = {}


def align_fwd(p_slabs, n_pairs, flags, jitter, extents, angles, want_stats=True):
    p = _np(p_slabs.sum(0))
    kw = {}
    crop, rotate, single = bool(flags & 1), bool(flags & 2), bool(flags & 4)
    if crop and not single:
        kw.update(jitter_x=np.concatenate([_np(jitter[0]), _np(jitter[1])]),
                  jitter_y=np.concatenate([_np(jitter[2]), _np(jitter[3])]))
    if rotate and not single:
        kw.update(angle=np.concatenate([_np(angles[0]), _np(angles[1])]))
    z, _, cache = O.align_fwd(p, n_pairs, crop=crop and not single, rotate=rotate and not single,
                              image_hw=(int(extents[0]), int(extents[1])), double_norm=not single, **kw)
    m = p.shape[0]
    n2 = cache["n2"] if cache["n2"] is not None else np.ones(m, np.float32)
    norms = torch.stack([_t(cache["n1"]), _t(n2.astype(np.float32))])
    row_stats = None
    if want_stats:
        pv = p.reshape(m, 64, 2)
        srt = np.sort(pv, axis=1)
        cols = []
        for c in range(2):
            cols += [pv[:, :, c].mean(1), srt[:, 31, c], pv[:, :, c].min(1), pv[:, :, c].max(1)]
        row_stats = _t(np.stack(cols, axis=1).astype(np.float32))
    zt = _t(z.astype(np.float32))
    _ALIGN_CACHE[zt.data_ptr()] = cache
    return _t(p), zt, norms, row_stats


def align_bwd(dz, p, z, norms, n_pairs, flags, angles):
    cache = _ALIGN_CACHE[z.data_ptr()]
    return _t(O.align_bwd(_np(dz), cache).astype(np.float32))


def ntxent_fwd(z_rows, row_offset, z_all, n_half, inv_tau, loss_scale, row_stats=None, n_pairs_stats=0,
               want_sim=False):
    z = _np(z_all)
    mr = z_rows.shape[0]
    _, s, lse, pos = O.ntxent_fwd(z, n_half, 1.0 / inv_tau)
    rows = slice(row_offset, row_offset + mr)
    out17 = torch.zeros(17)
    out17[16] = float((lse[rows] - pos[rows]).sum() * loss_scale)
    if row_stats is not None:
        out17[:16] = row_stats.view(2, n_pairs_stats, 8).mean(1).reshape(16)
    sim = _t(s[rows].astype(np.float32)) if want_sim else None
    return out17, _t(lse[rows].astype(np.float32)), sim


def ntxent_bwd(z_rows, row_offset, z_all, n_half, inv_tau, lse_all, dloss, grad_scale):
    mr, mg = z_rows.shape[0], z_all.shape[0]
    dz = O.ntxent_bwd(_np(z_all), _np(lse_all), n_half, 1.0 / inv_tau, dloss=float(dloss) * grad_scale * mg,
                      rows=slice(row_offset, row_offset + mr))
    return _t(dz.astype(np.float32))


def install(monkeypatch):
    from peclr_amd import _capi

    for name in ("pick_split_k", "gemm", "slab_reduce", "bn_relu_fwd", "bn_relu_bwd", "align_fwd", "align_bwd",
                 "ntxent_fwd", "ntxent_bwd"):
        monkeypatch.setattr(_capi, name, globals()[name])
