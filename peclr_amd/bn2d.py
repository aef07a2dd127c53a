"""Fused BatchNorm2d (+ residual add) (+ ReLU) for the NHWC backbone (kernels: csrc/bn2d.hip).

`FusedBatchNormAct2d` IS an `nn.BatchNorm2d` (same parameters, buffers and state_dict keys -- the
reference builds its ResNet with `norm_layer=nn.BatchNorm2d`, resnet_model.py:15) whose forward also
takes the block's residual and a ReLU flag, so a ResNet block can hand the whole
"bn -> (+identity) -> relu" tail to one fused HIP pass.

Two execution modes, chosen EXPLICITLY (no silent dispatch):
  hip = False (default)  stock PyTorch ops: F.batch_norm, add, relu -- any device / layout / dtype;
                         this is the PyTorch-ROCm backbone BASELINE.json:north_star describes.
  hip = True             the hand-written HIP kernels; requires fp32, bf16 or fp16 (autocast) channels_last
                         HIP tensors and raises otherwise.  Turn it on with `enable_hip_batchnorm`.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from . import _capi


# ---- activation checkpointing: a residual block keeps only its INPUT for the
# backward pass and re-runs its forward there.  While a block is being re-run, BatchNorm layers must not
# move their running statistics a second time (`_RECOMPUTING`); batch statistics are recomputed and are
# identical (the kernels are deterministic).
_RECOMPUTING = False


class _CheckpointedBlock(torch.autograd.Function):
    """y = run(x) without keeping run's intermediate activations; backward re-runs `run` on the saved input
    under the autocast state of the forward and back-propagates through that second graph (parameter
    gradients accumulate into `.grad` from inside, so gradient hooks -- the all-reduce buckets -- still fire)."""

    @staticmethod
    def forward(ctx, run, x):
        ctx.run = run
        ctx.autocast = (torch.is_autocast_enabled(x.device.type), torch.get_autocast_dtype(x.device.type))
        ctx.save_for_backward(x)
        with torch.no_grad():
            return run(x)

    @staticmethod
    def backward(ctx, dy):
        global _RECOMPUTING
        (x,) = ctx.saved_tensors
        x = x.detach().requires_grad_(True)
        enabled, dtype = ctx.autocast
        was = _RECOMPUTING
        _RECOMPUTING = True
        try:
            with torch.enable_grad(), torch.autocast(x.device.type, dtype=dtype, enabled=enabled):
                y = ctx.run(x)
        finally:
            _RECOMPUTING = was
        torch.autograd.backward((y,), (dy.to(y.dtype),))
        return None, x.grad


def checkpoint_block(run, x: Tensor) -> Tensor:
    """`run(x)` with activation checkpointing when a gradient will flow (otherwise plain)."""
    if torch.is_grad_enabled() and x.requires_grad and not _RECOMPUTING:
        return _CheckpointedBlock.apply(run, x)
    return run(x)


class _BN2dAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, bn: "FusedBatchNormAct2d", relu: bool):
        training = bn.training or not bn.track_running_stats
        # the ReLU mask can be recomputed from x unless a residual was added before it; then the
        # forward writes a 1-bit mask (or, for C % 32 != 0, the backward re-reads y)
        need_mask = relu and residual is not None
        rm, rv, nbt = bn._stat_buffers(training)
        y, save, ss, mask = _capi.bn2d_fwd(x, residual, weight, bias, rm, rv, nbt, training, bn.eps,
                                           bn.momentum if bn.momentum is not None else 0.1, relu, want_mask=need_mask,
                                           sync_group=bn.sync_group if training else None)
        keep = mask if mask is not None else (y if need_mask else None)
        ctx.save_for_backward(x, save, ss, *([keep] if keep is not None else []))
        ctx.cfg = (training, relu, residual is not None, keep is not None, mask is not None)
        ctx.sync_group = bn.sync_group if training else None
        return y

    @staticmethod
    def backward(ctx, dy):
        training, relu, has_res, has_keep, is_mask = ctx.cfg
        x, save, ss = ctx.saved_tensors[:3]
        keep = ctx.saved_tensors[3] if has_keep else None
        y, mask = (None, keep) if is_mask else (keep, None)
        dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx, dgamma, dbeta, dres = _capi.bn2d_bwd(dy, x, y, mask, save, ss, training, relu,
                                                 has_res and ctx.needs_input_grad[3], sync_group=ctx.sync_group)
        if has_res and dres is None and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dgamma, dbeta, dres, None, None


class _BN2dReluPool(torch.autograd.Function):
    """Stem: BatchNorm2d -> ReLU -> MaxPool2d(3, 2, 1) without ever writing the un-pooled activation."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn: "FusedBatchNormAct2d"):
        training = bn.training or not bn.track_running_stats
        sync = bn.sync_group if training else None
        rm, rv, nbt = bn._stat_buffers(training)
        y, x_at_max, code, save, ss = _capi.bn2d_pool_fwd(x, weight, bias, rm, rv, nbt, training, bn.eps,
                                                bn.momentum if bn.momentum is not None else 0.1, sync_group=sync)
        ctx.save_for_backward(x, x_at_max, code, save, ss)
        ctx.cfg = (training, sync)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x_at_max, code, save, ss = ctx.saved_tensors
        training, sync = ctx.cfg
        dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx, dgamma, dbeta = _capi.bn2d_pool_bwd(dy, x, x_at_max, code, save, ss, training, sync_group=sync)
        return dx, dgamma, dbeta, None


class _BN2dAddReluAvgPool(torch.autograd.Function):
    """Encoder tail (the last residual block's bn + identity + ReLU, then AdaptiveAvgPool2d((1, 1)) and
    flatten): returns the pooled fp32 [N, C] features; the [N, C, H, W] activation is never materialised,
    and the backward forms the pool's broadcast gradient on the fly."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, bn: "FusedBatchNormAct2d"):
        training = bn.training or not bn.track_running_stats
        sync = bn.sync_group if training else None
        rm, rv, nbt = bn._stat_buffers(training)
        pooled, mask, save, ss = _capi.bn2d_avgpool_fwd(x, residual, weight, bias, rm, rv, nbt, training, bn.eps,
                                                        bn.momentum if bn.momentum is not None else 0.1, sync_group=sync)
        ctx.save_for_backward(x, mask, save, ss)
        ctx.cfg = (training, sync)
        return pooled

    @staticmethod
    def backward(ctx, d_pooled):
        x, mask, save, ss = ctx.saved_tensors
        training, sync = ctx.cfg
        dx, dgamma, dbeta, dres = _capi.bn2d_avgpool_bwd(d_pooled.float().contiguous(), x, mask, save, ss, training,
                                                         sync_group=sync)
        return dx, dgamma, dbeta, dres, None


class _ForkConv1x1(torch.autograd.Function):
    """Bottleneck entry: (x, W) -> (conv1x1(x, W), x).  The block input feeds both the first convolution and
    the identity branch, so its gradient is dY W + d_identity: autograd runs MIOpen's dgrad and then an
    elementwise add over the block input (3.6 % of the fp32 step); here both are ONE fp32 GEMM with the
    identity gradient added in the epilogue.  Forward and the weight gradient stay on MIOpen."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.conv2d(x, weight), x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gid):
        x, weight = ctx.saved_tensors
        n, cin, h, w = x.shape
        cmid = weight.shape[0]
        dw = None
        gy = gy.to(x.dtype)
        if ctx.needs_input_grad[1]:
            dw = torch.ops.aten.convolution_backward(gy, x, weight.to(x.dtype), None, [1, 1], [0, 0], [1, 1], False,
                                                     [0, 0], 1, [False, True, False])[1].to(weight.dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            gy = gy.contiguous(memory_format=torch.channels_last)
            gid = gid.to(x.dtype).contiguous(memory_format=torch.channels_last)
            r = n * h * w
            a = gy.permute(0, 2, 3, 1).reshape(r, cmid)           # NHWC storage seen as [R, Cmid]: a view
            d = gid.permute(0, 2, 3, 1).reshape(r, cin)
            if x.dtype in (torch.bfloat16, torch.float16):       # autocast backbone: 16-bit MFMA, fp32 accumulate
                wt = weight.detach().reshape(cmid, cin).t().contiguous().to(x.dtype)
                out = _capi.gemm_add_half(a, wt, d, tag="conv1x1_dgrad_add")
            else:
                out = _capi.gemm_add(_capi.GEMM_NN, a, weight.reshape(cmid, cin), d, tag="conv1x1_dgrad_add")
            dx = out.view(n, h, w, cin).permute(0, 3, 1, 2)       # back to a channels_last NCHW tensor
        return dx, dw


def fork_conv1x1(conv: nn.Conv2d, x: Tensor):
    """`(conv(x), x)` with the fused input gradient when `conv.hip_fork` is set (enable_hip_batchnorm does
    it for the bottlenecks' first 1x1 convolution) and the activations are channels_last on a HIP device,
    fp32 (no autocast) or bf16 (under bf16 autocast); the stock ops otherwise."""
    autocast = torch.is_autocast_enabled("cuda")
    dtype_ok = ((x.dtype == torch.float32 and not autocast)
                or (x.dtype in (torch.bfloat16, torch.float16) and autocast and torch.get_autocast_dtype("cuda") == x.dtype))
    ok = (getattr(conv, "hip_fork", False) and x.is_cuda and dtype_ok
          and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
          and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and x.requires_grad
          and x.is_contiguous(memory_format=torch.channels_last)
          and conv.weight.shape[0] % 8 == 0 and conv.weight.shape[1] % 8 == 0)
    if ok:
        return _ForkConv1x1.apply(x, conv.weight)
    return conv(x), x


class FusedBatchNormAct2d(nn.BatchNorm2d):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, **kw):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine,
                         track_running_stats=track_running_stats, **kw)
        self.hip = False
        self.default_relu = False  # stem BN inside an nn.Sequential: fuse the ReLU that follows it
        self.default_pool = False  # ... and the MaxPool2d(3, 2, 1) after that (the encoder's stem)
        # last BatchNorm of layer4 inside the encoder wrapper: in HIP mode its fused pass also performs the
        # AdaptiveAvgPool2d((1, 1)) + flatten that follow the block and returns fp32 [N, C]
        self.tail_avgpool = False
        # data parallel: a process group -> training statistics over the rows of ALL its ranks (what one
        # device holding the concatenated batch computes); None -> per-rank statistics (DDP's default)
        self.sync_group = None

    def _stat_buffers(self, training: bool):
        """Running statistics the kernels should update -- none while a checkpointed block is being re-run in
        the backward pass (the first run already moved them), unless they are needed as the common shift of
        synchronised statistics."""
        if training and _RECOMPUTING and self.sync_group is None:
            return None, None, None
        return self.running_mean, self.running_var, self.num_batches_tracked

    def forward(self, x: Tensor, residual: Optional[Tensor] = None, relu: Optional[bool] = None) -> Tensor:
        relu = self.default_relu if relu is None else relu
        pool = self.default_pool and relu and residual is None
        if self.hip:
            if not self.affine or (self.training and self.momentum is None):
                raise _capi.PeclrHipError("fused BatchNorm2d needs affine=True and a fixed momentum")
            if pool:
                return _BN2dReluPool.apply(x, self.weight, self.bias, self)
            if self.tail_avgpool and relu and residual is not None and self.num_features % 32 == 0:
                return _BN2dAddReluAvgPool.apply(x, self.weight, self.bias, residual, self)
            return _BN2dAct.apply(x, self.weight, self.bias, residual, self, relu)
        if self.sync_group is not None and self.training:
            raise _capi.PeclrHipError("synchronised statistics are implemented by the HIP kernels only (hip=True)")
        if self.training and _RECOMPUTING:   # re-run of a checkpointed block: batch statistics, buffers untouched
            y = F.batch_norm(x, None, None, self.weight, self.bias, True, 0.0, self.eps)
        else:
            y = super().forward(x)
        if residual is not None:
            y = y + residual
        y = F.relu(y) if relu else y
        return F.max_pool2d(y, 3, stride=2, padding=1) if pool else y


def enable_hip_batchnorm(module: nn.Module, enabled: bool = True, sync_group=None) -> int:
    """Switch every FusedBatchNormAct2d under `module` to the HIP kernels (or back).  The caller is
    responsible for feeding fp32 channels_last HIP tensors; anything else raises.
    sync_group: process group whose ranks share their batch statistics (None = per-rank)."""
    n = 0
    for m in module.modules():
        if isinstance(m, FusedBatchNormAct2d):
            m.hip = enabled
            m.sync_group = sync_group if enabled else None
            n += 1
        elif getattr(m, "fork_entry", False):   # bottleneck conv1 (resnet.Bottleneck marks it)
            m.hip_fork = enabled
    return n
