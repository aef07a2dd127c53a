"""Autograd wrappers around the HIP kernels: the two differentiable operators of the hot path.

  head_align(h, head params, aug params) -> z, row_stats        K1..K7  (+ closed-form backward)
  ntxent(z_local, ...)                   -> loss, stats16       K8      (+ closed-form backward)

Both call libpeclr_hip.so through `_capi` and nothing else: no torch fallback, no CPU path.  The
split between the two is where data parallelism cuts the path: `ntxent` all-gathers the projected
embeddings (and the per-row log-denominators) so every rank sees the full negative set
(SURVEY.md section 8e); with a single process it degenerates to the reference's loss.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _capi
from . import dist as pdist


@dataclass
class AlignSpec:
    """What `Hybrid2Model.get_transformed_projections` reads from the batch and the config
    (hybrid2_model.py:58-80).  Tensors are used exactly as the batch dict holds them."""
    n_pairs: int
    crop: bool = False
    rotate: bool = False
    single_norm: bool = False          # SimCLR.contrastive_step (simclr_model.py:44-47)
    jitter: Optional[Tuple[Tensor, Tensor, Tensor, Tensor]] = None  # jx1, jx2, jy1, jy2 (int64)
    extents: Tuple[float, float] = (1.0, 1.0)                       # float(shape[0]), float(shape[1])
    angles: Optional[Tuple[Tensor, Tensor]] = None                  # angle_1, angle_2 (float64)
    want_stats: bool = True

    @property
    def flags(self) -> int:
        if self.single_norm:
            return _capi.ALIGN_SINGLE_NORM
        return (_capi.ALIGN_CROP if self.crop else 0) | (_capi.ALIGN_ROTATE if self.rotate else 0)


@dataclass
class BNState:
    training: bool
    eps: float
    momentum: float
    running_mean: Optional[Tensor]
    running_var: Optional[Tensor]
    num_batches_tracked: Optional[Tensor]
    sync_group: object = None  # process group sharing the batch statistics (None = per-rank)


_REGISTER_BN_MAX_ROWS = 1024  # bn_relu kernels hold <= 16 rows per thread x 64 row slices in registers


def _as_slabs(t: Tensor) -> Tensor:
    return t if t.dim() == 3 else t.unsqueeze(0)


class _HeadAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, w1, b1, gamma, beta, w2, bn: BNState, spec: AlignSpec):
        h = h.contiguous()
        m, din = h.shape
        hid, d = w1.shape[0], w2.shape[0]
        ss = None
        sync = bn.sync_group if bn.training else None
        if m <= _REGISTER_BN_MAX_ROWS and sync is None:
            # K1: a = h W1^T  (split-K slabs; bias and the slab reduction are fused into the BN kernel,
            # which keeps its rows in registers)
            a_slabs = _as_slabs(_capi.gemm(_capi.GEMM_NT, h, w1, split_k=_capi.pick_split_k(m, hid, din),
                                           tag="gemm_k1_fwd"))
            a_pre, a, save = _capi.bn_relu_fwd(a_slabs, b1, gamma, beta, bn.eps, bn.momentum, bn.training,
                                               bn.running_mean, bn.running_var, bn.num_batches_tracked)
        else:
            # large batches: enough tiles without split-K (bias in the GEMM epilogue), and the [M, H]
            # activation matrix IS an NHWC tensor with 1x1 spatial extent -> the streaming
            # stats / finalize / apply kernels of the backbone glue (4.5-5.5 TB/s) instead of the
            # register-resident kernel, which only scales to 1024 rows.  Synchronised statistics
            # (all-reduce between stats and finalize) also take this three-phase route.
            a_pre = _capi.gemm(_capi.GEMM_NT, h, w1, bias=b1, tag="gemm_k1_fwd")
            a4, save, ss, _ = _capi.bn2d_fwd(a_pre.view(m, hid, 1, 1), None, gamma, beta, bn.running_mean,
                                            bn.running_var, bn.num_batches_tracked, bn.training, bn.eps,
                                            bn.momentum, relu=True, sync_group=sync)
            a = a4.view(m, hid)
        # K2: p = relu(bn(a)) W2^T  (slabs reduced inside the align kernel)
        p_slabs = _as_slabs(_capi.gemm(_capi.GEMM_NT, a, w2, split_k=_capi.pick_split_k(m, d, hid), tag="gemm_k2_fwd"))
        p, z, norms, row_stats = _capi.align_fwd(p_slabs, spec.n_pairs, spec.flags, spec.jitter, spec.extents,
                                                 spec.angles, spec.want_stats)
        ctx.save_for_backward(h, w1, gamma, beta, w2, a_pre, a, save, p, z, norms, *([ss] if ss is not None else []))
        ctx.spec, ctx.bn_training, ctx.sync_group = spec, bn.training, sync
        if row_stats is None:
            row_stats = torch.empty(0, device=h.device)
        ctx.mark_non_differentiable(row_stats)
        return z, row_stats

    @staticmethod
    def backward(ctx, dz, _d_stats):
        h, w1, gamma, beta, w2, a_pre, a, save, p, z, norms = ctx.saved_tensors[:11]
        ss = ctx.saved_tensors[11] if len(ctx.saved_tensors) > 11 else None
        spec = ctx.spec
        dp = _capi.align_bwd(dz.contiguous(), p, z, norms, spec.n_pairs, spec.flags, spec.angles)
        dw2 = _capi.gemm(_capi.GEMM_TN, dp, a, tag="gemm_dw2")            # [D,H]   = dp^T a
        da = _capi.gemm(_capi.GEMM_NN, dp, w2, tag="gemm_da")            # [M,H]   = dp W2
        if ss is None:
            d_a_pre, dgamma, dbeta, db1 = _capi.bn_relu_bwd(da, a_pre, save, gamma, beta, ctx.bn_training)
        else:
            m, hid = a_pre.shape
            dx4, dgamma, dbeta, _ = _capi.bn2d_bwd(da.view(m, hid, 1, 1), a_pre.view(m, hid, 1, 1), None, None, save,
                                                   ss, ctx.bn_training, True, False, sync_group=ctx.sync_group)
            d_a_pre = dx4.view(m, hid)
            # column sum of d_a_pre in closed form: 0 through batch statistics (with synchronised
            # statistics: 0 once summed over the ranks, which the gradient all-reduce does),
            # scale*dbeta through frozen ones
            db1 = torch.zeros_like(dbeta) if ctx.bn_training else ss[0] * dbeta
        dw1 = _capi.gemm(_capi.GEMM_TN, d_a_pre, h, tag="gemm_dw1")       # [H,Din] = dA^T h
        dh = _capi.gemm(_capi.GEMM_NN, d_a_pre, w1, tag="gemm_dh") if ctx.needs_input_grad[0] else None
        return dh, dw1, db1, dgamma, dbeta, dw2, None, None


def head_align(h: Tensor, w1: Tensor, b1: Tensor, gamma: Tensor, beta: Tensor, w2: Tensor, bn: BNState,
               spec: AlignSpec) -> Tuple[Tensor, Tensor]:
    """Projection head + equivariance alignment.  Returns (z [M,128] unit rows, row_stats [M,8])."""
    if h.dtype != torch.float32:
        h = h.float()  # bf16/fp16 autocast backbones: the head, logits and loss stay fp32
    return _HeadAlign.apply(h, w1, b1, gamma, beta, w2, bn, spec)


class _NTXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_local, row_stats, n_pairs, temperature, group, want_sim):
        z_local = z_local.contiguous()
        mr = z_local.shape[0]
        world = pdist.world_size(group)
        rank = pdist.rank(group)
        multi = pdist.collectives_active(group)
        z_all = pdist.all_gather_cat(z_local, group) if multi else z_local
        mg = z_all.shape[0]
        stats_in = row_stats if (row_stats is not None and row_stats.numel() > 0) else None
        out17, row_lse, sim = _capi.ntxent_fwd(z_local, rank * mr, z_all, n_pairs, 1.0 / temperature, 1.0 / mg,
                                               stats_in, n_pairs, want_sim)
        if multi:
            # one collective carries every rank's log-denominators, its partial loss AND its 16 projection
            # statistics (means over this rank's samples; equal pair counts per rank, so the mean of the
            # rank means is the global-batch mean a single device would report)
            packed = pdist.all_gather_cat(torch.cat([row_lse, out17]), group).view(world, mr + 17)
            lse_all = packed[:, :mr].reshape(-1).contiguous()
            loss = packed[:, mr + 16].sum()
            out17 = torch.cat([packed[:, mr:mr + 16].mean(dim=0), loss.reshape(1)])
        else:
            lse_all, loss = row_lse, out17[16].clone()
        ctx.save_for_backward(z_local, z_all, lse_all)
        ctx.meta = (rank * mr, n_pairs, 1.0 / temperature, 1.0 / mg)
        if sim is None:
            sim = torch.empty(0, device=z_local.device)
        ctx.mark_non_differentiable(out17, sim)
        return loss, out17, sim

    @staticmethod
    def backward(ctx, dloss, _ds, _dsim):
        z_local, z_all, lse_all = ctx.saved_tensors
        row_offset, n_half, inv_tau, grad_scale = ctx.meta
        dloss = dloss.detach().reshape(1).to(torch.float32).contiguous()
        dz = _capi.ntxent_bwd(z_local, row_offset, z_all, n_half, inv_tau, lse_all, dloss, grad_scale)
        return dz, None, None, None, None, None


def ntxent(z_local: Tensor, n_pairs: int, temperature: float = 0.5, row_stats: Optional[Tensor] = None,
           group=None, want_sim: bool = False):
    """NT-Xent over the GLOBAL batch (vanila_contrastive_loss semantics, utils.py:154-186).

    z_local: [2*n_pairs, 128] unit rows of this rank (view-1 rows then view-2 rows).
    Returns (loss, stats16, sim): loss is the mean over all world*2*n_pairs rows and is identical on
    every rank; its gradient w.r.t. z_local is d(loss_global)/d(z_local), so parameter gradients
    must be SUMMED over ranks (peclr_amd.dist.GradReducer does that).  stats16 = means of row_stats
    over the GLOBAL batch (all ranks' samples; every rank must hold n_pairs pairs -- the Trainer checks
    that).  sim = [Mr, Mg] similarities when want_sim.
    """
    loss, out17, sim = _NTXent.apply(z_local, row_stats, n_pairs, temperature, group, want_sim)
    return loss, out17[:16], sim
