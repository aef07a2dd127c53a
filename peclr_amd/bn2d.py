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

import contextlib
import dataclasses
import os

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
_FIRST_RUN = False      # inside the no-grad first run of a checkpointed block (its BatchNorm layers note the statistics' shift)
# The statistics' shifts of the checkpointed block that is running: id(BatchNorm layer) -> the copy of its running mean the
# first run centred its sums on.  The dictionary belongs to ONE invocation (it lives on that invocation's autograd context):
# the first run fills it, the re-run of the same invocation reads it back -- whatever other forward passes (a second view
# pass, a no-grad evaluation, another accumulation micro-batch) went through the same layers in between.
_CKPT_SHIFTS = None


def _ckpt_shift_of(bn):
    """First run of a checkpointed block: take (once) and return the copy of `bn`'s running mean that this invocation centres
    its statistics on; re-run: the same copy; outside a checkpointed block: None."""
    if _CKPT_SHIFTS is None or not (_FIRST_RUN or _RECOMPUTING):
        return None
    key = id(bn)
    if _FIRST_RUN and key not in _CKPT_SHIFTS:
        _CKPT_SHIFTS[key] = bn.running_mean.detach().clone()
    return _CKPT_SHIFTS.get(key)


class _CheckpointedBlock(torch.autograd.Function):
    """y = run(x) without keeping run's intermediate activations; backward re-runs `run` on the saved input
    under the autocast state of the forward and back-propagates through that second graph (parameter
    gradients accumulate into `.grad` from inside, so gradient hooks -- the all-reduce buckets -- still fire)."""

    @staticmethod
    def forward(ctx, run, x):
        ctx.run = run
        ctx.autocast = (torch.is_autocast_enabled(x.device.type), torch.get_autocast_dtype(x.device.type))
        ctx.save_for_backward(x)
        ctx.shifts = {}
        global _FIRST_RUN, _CKPT_SHIFTS
        was = (_FIRST_RUN, _CKPT_SHIFTS)
        _FIRST_RUN, _CKPT_SHIFTS = True, ctx.shifts
        try:
            with torch.no_grad():
                return run(x)
        finally:
            _FIRST_RUN, _CKPT_SHIFTS = was

    @staticmethod
    def backward(ctx, dy):
        global _RECOMPUTING, _CKPT_SHIFTS
        (x,) = ctx.saved_tensors
        x = x.detach().requires_grad_(True)
        enabled, dtype = ctx.autocast
        was = (_RECOMPUTING, _CKPT_SHIFTS)
        _RECOMPUTING, _CKPT_SHIFTS = True, ctx.shifts
        try:
            with torch.enable_grad(), torch.autocast(x.device.type, dtype=dtype, enabled=enabled):
                y = ctx.run(x)
        finally:
            _RECOMPUTING, _CKPT_SHIFTS = was
        # (ctx.shifts stays with the context: a second backward over a retained graph re-runs the block centred where the first
        # run was and reproduces it bit for bit; the copies -- one running mean per BatchNorm layer of the block -- go with the graph)
        global _NESTED_BACKWARD
        _NESTED_BACKWARD += 1       # (its side-channel entries are drained with the enclosing pass: see _drain_after_backward)
        try:
            torch.autograd.backward((y,), (dy.to(y.dtype),))
        finally:
            _NESTED_BACKWARD -= 1
        _drain_after_backward()
        return None, x.grad


def checkpoint_block(run, x: Tensor) -> Tensor:
    """`run(x)` with activation checkpointing when a gradient will flow (otherwise plain)."""
    if torch.is_grad_enabled() and x.requires_grad and not _RECOMPUTING:
        return _CheckpointedBlock.apply(run, x)
    return run(x)




# gradient tensors whose producer (an input-gradient GEMM) already reduced them against the BatchNorm layer they arrive at:
# data_ptr -> (token of that layer's forward, partial sums, n_split); popped by the layer's backward
_BN_BWD_STATS: dict = {}

def _env_flag(name: str, default: str = "1") -> bool:
    return os.environ.get(name, default) != "0"


@dataclasses.dataclass
class Routing:
    """Every switch that decides which kernel a layer of the fused backbone runs on, in ONE object (`bn2d.ROUTING`).
    The defaults come from the `PECLR_*` environment variables (same-box A/B runs of one build); tests and tools change
    them through `with bn2d.routing(name=value, ...)`.  `force`: route every shape the in-tree kernels ACCEPT to them,
    ignoring the "does it beat MIOpen at this size" tests below -- so that a small problem exercises the same kernels, in
    the same composition, as the full-size configurations (the kernels' own shape constraints still apply)."""
    gemm_x6: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_GEMM_X6"))              # fp32 GEMM-shaped work as six bf16 MFMA products
    x6_min_k: int = dataclasses.field(default_factory=lambda: int(os.environ.get("PECLR_GEMM_X6_MIN_K", "128")))   # in-step A/B: 128 beats 256 by 0.1-0.4 ms
    gemm_x6p: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_GEMM_X6P"))            # weight planes packed once per step (peclr_gemm_x6p_f32)
    gemm_x6t: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_GEMM_X6T"))            # weight gradients on the 256 x 256-tile kernel
    bn_stats_in_gemm: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_BN_STATS_IN_GEMM"))   # BatchNorm statistics in the GEMM epilogue
    bn_bwd_in_gemm: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_BN_BWD_IN_GEMM"))       # BatchNorm backward reduction in the dgrad epilogue
    # the shortcut's BatchNorm (conv1x1 -> bn of a layer's first block) applied inside the block's last pass (bn3 + shortcut + ReLU)
    bn_shortcut_in_add: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_BN_SHORTCUT_IN_ADD"))
    x6_layer1: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_X6_LAYER1"))          # layer1's 64-channel 1x1 convolutions in-tree
    x6_layer1_fork: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_X6_LAYER1_FORK"))   # layer1's fused entry gradient (K = 64)
    x6_layer1_wgrad: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_X6_LAYER1_WGRAD"))  # layer1's 64-wide weight gradients
    conv3x3_x6: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_CONV3X3_X6"))        # 3x3 stride-1 convolutions as implicit GEMMs
    conv3x3_wgrad_x6: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_CONV3X3_WGRAD_X6"))   # their weight gradients (nine taps, one launch)
    conv3x3_tile_rows: int = dataclasses.field(default_factory=lambda: int(os.environ.get("PECLR_CONV3X3_TILE_ROWS", "256")))
    # (pair arithmetic: the rounds-of-slots policy, 0 -- same-box A/B at C2: 47.53 ms per step against 47.73 / 47.94 with 128 / 256 rows)
    conv3x3_pair_tile_rows: int = dataclasses.field(default_factory=lambda: int(os.environ.get("PECLR_CONV3X3_PAIR_TILE_ROWS", "0")))
    conv_s2_x6: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_CONV_S2_X6"))        # forward of the stride-2 convolutions
    conv_s2_wgrad_x6: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_CONV_S2_WGRAD_X6"))
    conv_s2_dgrad_x6: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_CONV_S2_DGRAD_X6"))   # 3x3 / stride-2 input gradient by parity classes
    s2_tile_rows: int = dataclasses.field(default_factory=lambda: int(os.environ.get("PECLR_CONV_S2_TILE_ROWS", "0")))   # 0 = the rounds-of-slots policy
    s2_dgrad_compact: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_S2_DGRAD_COMPACT"))   # the shortcut's compact input gradient
    lazy_residual_grad: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_LAZY_RESIDUAL_GRAD"))   # identity shortcut: (dy, mask) hand-over
    conv16: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_CONV16"))                # 16-bit (bf16 / fp16 autocast) convolutions in-tree
    # fp32 forward / input-gradient GEMMs in "pair" arithmetic (round 6): operands scaled by a per-tensor power of two and split into
    # two fp16 numbers, three MFMA products instead of six -- where the operand tensor carries its maximum (`_peclr_absmax`, written
    # by the BatchNorm pass that produced it); anything else runs the six-product kernels as before
    x6_pair: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_X6_PAIR"))
    wgrad16: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_WGRAD16"))              # ... and their weight gradients
    stem_wgrad: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_STEM_WGRAD"))        # ... and its weight gradient
    stem: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_STEM"))                    # the 7x7 / stride-2 stem forward in-tree (csrc/stem.hip)
    force: bool = dataclasses.field(default_factory=lambda: _env_flag("PECLR_ROUTE_FORCE", "0"))

    # ---- "does the in-tree kernel pay at this shape" (measured on ResNet-50's shapes; the kernels accept far more)
    def tiles(self, rows: int, n_out: int) -> int:
        """Workgroup tiles of a [rows, n_out] output on 128-row x 128-column (64 for narrow outputs) tiles."""
        return (rows // 128) * max(1, n_out // (128 if n_out >= 128 else 64))

    MALL_BYTES = 256 << 20        # Infinity Cache (MI355X_MICROARCH.md): tensors beyond it stream from HBM

    def streams_from_hbm(self, rows: int, k: int, n_out: int) -> bool:
        """The product's fp32 operand + output rows exceed the Infinity Cache: a separate BatchNorm statistics /
        backward-reduction pass over them would come from HBM again.  This is what makes the in-tree GEMM pay for the
        64-channel shapes (layer1), where the GEMM itself is a draw against MIOpen (both HBM-bound: tools/exp/layer1_probe.py)
        and the fused epilogue saves that pass."""
        return self.force or 4 * rows * (k + n_out) >= self.MALL_BYTES

    def fills_chip(self, rows: int, n_out: int, rounds: float = 0.765) -> bool:
        """At least `rounds` of the chip's 256 CUs get a tile: below that the launch is latency- and tail-bound and
        MIOpen's smaller tiles win (ResNet-50 @224 with 2 x 128 views: every routed shape has >= 196 tiles)."""
        return self.force or self.tiles(rows, n_out) >= 256 * rounds


ROUTING = Routing()


@contextlib.contextmanager
def routing(**overrides):
    """Temporarily change fields of `ROUTING` (tests, A/B tools): `with bn2d.routing(bn_bwd_in_gemm=False): ...`."""
    old = {k: getattr(ROUTING, k) for k in overrides}
    for k, v in overrides.items():
        if not hasattr(ROUTING, k):
            raise AttributeError(f"Routing has no field {k!r}")
        setattr(ROUTING, k, v)
    try:
        yield ROUTING
    finally:
        for k, v in old.items():
            setattr(ROUTING, k, v)


class X6PackGroup:
    """The 1x1-convolution weights of one encoder that run as peclr_gemm_x6p_f32 GEMMs, split into fragment-ordered
    bf16 planes by ONE launch per optimiser step (peclr_x6_pack_f32): W[Cout][Cin] for the forward GEMM, W^T for the
    input-gradient GEMMs.  A member convolution asks `planes(conv)` right before it launches; the group re-packs
    everything when that convolution's weight changed since the last pack (in-place update: `_version`; fused optimiser
    step, which writes through raw pointers: `_capi.WEIGHTS_EPOCH`; new storage: `data_ptr`).  While a hipGraph is
    being captured, the FIRST `planes()` call of that capture (whichever member makes it, whatever the stamps say) packs:
    the host-side stamps describe the moment of recording, a replay must re-split the weights it is about to use."""

    def __init__(self, convs):
        self.convs = [c for c in convs if self.member(c)]
        # one plane set per operand format: None = the three bf16 planes of the fp32 six-product kernels (peclr_x6_pack_f32);
        # torch.bfloat16 / torch.float16 = the single 16-bit plane of the autocast kernels (peclr_h_pack), packed from the same
        # fp32 master weights.  Each: [planes object, parameter addresses, stamps, capture that already holds a pack launch]
        self._sets = {}
        for i, c in enumerate(self.convs):   # (position kept on the module: a deep copy of the model keeps group and members consistent)
            c.x6_group, c.x6_index = self, i

    @staticmethod
    def member(conv) -> bool:
        cout, cin = conv.out_channels, conv.in_channels
        if conv.stride not in ((1, 1), (2, 2)) or conv.groups != 1 or conv.bias is not None:
            return False            # (stride 2: the forward only, `_ConvS2Gemm`)
        if conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.dilation == (1, 1):
            return cout >= 64 and cin >= 64 and cout % 64 == 0 and cin % 64 == 0        # 64-column tiles for layer1
        return (conv.kernel_size == (1, 1) and conv.padding == (0, 0) and cout >= 64 and cin >= 64
                and cout % 64 == 0 and cin % 64 == 0)

    def _key(self, conv):
        return (conv.weight.data_ptr(), conv.weight._version, _capi.WEIGHTS_EPOCH)

    def _specs(self):
        specs = []
        for c in self.convs:
            if c.kernel_size == (3, 3):
                # [Cout][3][3][Cin] as it lies in (channels_last) memory: forward B_t = [Cout, 9 Cin]; input gradient
                # B_t[ci][tap * Cout + co] = W[co][tap][ci]
                w4 = c.weight.detach().permute(0, 2, 3, 1)
                if not w4.is_contiguous():
                    raise _capi.PeclrHipError("X6PackGroup: 3x3 weights must be channels_last (NHWC encoder)")
                specs += [(w4.reshape(c.out_channels, 9 * c.in_channels), False), (w4.reshape(c.out_channels * 9, c.in_channels), 9)]
            else:
                w2 = c.weight.detach().reshape(c.out_channels, c.in_channels)
                specs += [(w2, False), (w2, True)]
        return specs

    def pack(self, dtype=None):
        st = self._sets.setdefault(dtype, [None, None, [None] * len(self.convs), 0])
        ptrs = [c.weight.data_ptr() for c in self.convs]
        if st[0] is None or ptrs != st[1]:
            if _capi.capture_id() != 0:
                # building the plane set uploads its pointer table (a host-to-device copy) and allocates the planes: neither
                # may be recorded into a hipGraph
                raise _capi.PeclrHipError("X6PackGroup: the weight planes of this precision do not exist yet and cannot be created "
                                          "while a hipGraph is being captured -- run one eager forward pass (same autocast dtype) first")
            st[0] = (_capi.X6Planes(self._specs()) if dtype is None else _capi.X6Planes(self._specs(), pair=True) if dtype == "pair"
                     else _capi.HPlanes(self._specs(), dtype))
            st[1] = ptrs
        st[0].pack()
        st[2] = [self._key(c) for c in self.convs]

    def planes(self, conv, dtype=None):
        """(forward planes of W [Cout, Cin], input-gradient planes of W^T), fresh; dtype None: the fp32 kernels' three-plane
        format, "pair": their fp16-pair format (`pair_scales` gives the matrices' powers of two), torch.bfloat16 / torch.float16:
        the 16-bit kernels' format."""
        at = conv.x6_index
        if at >= len(self.convs) or self.convs[at] is not conv:
            raise _capi.PeclrHipError("X6PackGroup: convolution is not a member of its group (call enable_hip_batchnorm again)")
        st = self._sets.setdefault(dtype, [None, None, [None] * len(self.convs), 0])
        stale = st[2][at] != self._key(conv)
        cap = _capi.capture_id()
        if cap != st[3]:          # a new capture (or back to eager launches): this capture has no pack launch yet
            stale = stale or cap != 0
            st[3] = cap
        if stale:
            self.pack(dtype)
        return st[0].planes[2 * at], st[0].planes[2 * at + 1]

    def pair_scales(self, conv):
        """The device floats holding the powers of two of `conv`'s two pair-format matrices (after `planes(conv, "pair")`)."""
        st = self._sets["pair"][0]
        return st.scale(2 * conv.x6_index), st.scale(2 * conv.x6_index + 1)


def _x6_planes(conv):
    """Packed planes of `conv`'s weight, or None when the convolution is not in a pack group (then the GEMMs split the
    weight per workgroup: peclr_gemm_x6_f32)."""
    group = getattr(conv, "x6_group", None) if conv is not None else None
    if group is None or not ROUTING.gemm_x6p or not conv.weight.is_cuda or conv.weight.dtype != torch.float32:
        return None
    return _LazyPlanes(group, conv)


class _LazyPlanes:
    """`planes[0]` / `planes[1]` of a convolution's six-product plane set, asked of the group (which re-packs a stale set: one launch
    for all members) only when a launch really takes them -- with `ROUTING.x6_pair` most launches take the pair set instead, and
    a step in which nothing falls back never packs the three-plane set at all."""

    __slots__ = ("group", "conv")

    def __init__(self, group, conv):
        self.group, self.conv = group, conv

    def __getitem__(self, which):
        return self.group.planes(self.conv)[which]


def _absmax_of(t):
    """The device float holding max |t| if the pass that wrote `t` left one (fp32 tensors, `ROUTING.x6_pair`) AND nothing has
    written to `t` since (its version counter stands where the pass left it: an in-place update would make the maximum -- and with
    it the fp16 range the pair kernels scale into -- stale), else None."""
    if not ROUTING.x6_pair or t is None:
        return None
    tag = getattr(t, "_peclr_absmax", None)
    return tag[0] if (tag is not None and tag[1] == t._version) else None


def _tag_absmax(t: Tensor, slot):
    """Attach the maximum `slot` (a one-element device tensor) to the tensor it describes."""
    t._peclr_absmax = (slot, t._version)


def _new_absmax(x: Tensor):
    """A zeroed slot for the maximum of a tensor a BatchNorm pass is about to write (None: not wanted)."""
    return _capi.absmax_slot(x.device) if (ROUTING.x6_pair and ROUTING.gemm_x6p and x.is_cuda and x.dtype == torch.float32) else None


def _pair_planes(conv, amax, which: int, planes):
    """What a packed-weight GEMM of `conv` runs on: (planes, {"pair": (amax, w_scale)}) -- the fp16-pair planes of matrix `which`
    (0 forward, 1 input gradient) when the activation operand's maximum `amax` is known -- else (planes[which], {}): the
    six-product arithmetic."""
    group = getattr(conv, "x6_group", None) if conv is not None else None
    if amax is None or group is None or not ROUTING.x6_pair:
        return planes[which], {}
    pp = group.planes(conv, "pair")
    return pp[which], {"pair": (amax, group.pair_scales(conv)[which])}


_HALF = (torch.bfloat16, torch.float16)


def _h_ok(conv, x: Tensor) -> bool:
    """Does `conv` run on the in-tree 16-bit kernels for input `x`?  bf16 / fp16 NHWC activations under the matching autocast,
    fp32 master weights in a pack group (peclr_h_pack packs them once per step: autocast's per-forward cast is gone)."""
    return (ROUTING.conv16 and x.is_cuda and x.dtype in _HALF and x.dim() == 4 and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == x.dtype and getattr(conv, "x6_group", None) is not None
            and conv.weight.is_cuda and conv.weight.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last))


class _BN2dAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, bn: "FusedBatchNormAct2d", relu: bool, pre=None, link=None, lazy_res=False, defer=None,
                res_deferred=None, amax=None):
        """link: None or an empty list that receives what a consumer's input-gradient GEMM needs to perform this layer's
        backward reduction in its epilogue: [x, save, scale_shift, relu mask or None, relu, token].
        defer: None, or an empty list -- the layer (a shortcut's BatchNorm, no ReLU) only finishes its statistics; the list comes
        back as [x, scale_shift, relu] and the result is a placeholder (`_deferred_view`) whose consumer -- the block's last
        BatchNorm pass -- computes the layer on the fly; any other reader materialises it (`_Materialize`).
        res_deferred: None, or (x_s, scale_shift_s, False) -- `residual` is the placeholder of the shortcut's BatchNorm layer (no
        ReLU), which left its apply pass to THIS pass: the residual is computed from that layer's input on the fly.
        amax: None, or an empty list that receives the device float holding max |y| (the "pair" GEMMs' operand scale)."""
        training = bn.training or not bn.track_running_stats
        # the ReLU mask can be recomputed from x unless a residual was added before it; then the
        # forward writes a 1-bit mask (or, for C % 32 != 0, the backward re-reads y)
        need_mask = relu and residual is not None
        rm, rv, nbt, shift = bn._stat_buffers(training)
        slot = _new_absmax(x) if (amax is not None and defer is None) else None
        y, save, ss, mask = _capi.bn2d_fwd(x, None if res_deferred is not None else residual, weight, bias, rm, rv, nbt, training, bn.eps,
                                           bn.momentum if bn.momentum is not None else 0.1, relu, want_mask=need_mask,
                                           sync_group=bn.sync_group if training else None, sync_shift=shift,
                                           pre=pre if training else None, apply=defer is None,
                                           residual_bn=res_deferred[:2] if res_deferred is not None else None, absmax=slot)
        if slot is not None:
            amax[:] = [slot]
        if defer is not None:
            defer[:] = [x, ss, relu]
            y = _deferred_view(x)
        keep = mask if mask is not None else (y if need_mask else None)
        ctx.save_for_backward(x, save, ss, *([keep] if keep is not None else []))
        ctx.cfg = (training, relu, residual is not None, keep is not None, mask is not None)
        ctx.sync_group = bn.sync_group if training else None
        ctx.token = None
        # the residual's gradient is relu'(y) * dy: when its only consumer is the block's entry-gradient GEMM (which then
        # reads dy and the 1-bit mask itself), it is handed over as that pair instead of being written out
        ctx.lazy_res = bool(lazy_res and mask is not None and (x.dtype == torch.float32 or ROUTING.conv16))
        if link is not None and (x.dtype == torch.float32 or ROUTING.conv16) and (keep is None or mask is not None):
            ctx.token = object()
            link[:] = [x, save, ss, mask, relu, ctx.token]
        return y

    @staticmethod
    def backward(ctx, dy):
        training, relu, has_res, has_keep, is_mask = ctx.cfg
        x, save, ss = ctx.saved_tensors[:3]
        keep = ctx.saved_tensors[3] if has_keep else None
        y, mask = (None, keep) if is_mask else (keep, None)
        dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        pre = None
        ent = _BN_BWD_STATS.pop(dy.data_ptr(), None) if ctx.token is not None else None
        # the GEMM that produced dy reduced it against x already -- valid only for the tensor that GEMM wrote, untouched:
        # a second consumer of this layer's output makes autograd add its gradient INTO that buffer (in place: the
        # version counter moves), and the sums would miss that contribution
        if ent is not None and ent[0] is ctx.token and ent[3] == dy._version:
            pre = ent[1:3]
        lazy = ctx.lazy_res and has_res and ctx.needs_input_grad[3] and ROUTING.lazy_residual_grad and not torch.is_anomaly_enabled()
        slot = _new_absmax(x)
        dx, dgamma, dbeta, dres = _capi.bn2d_bwd(dy, x, y, mask, save, ss, training, relu,
                                                 has_res and ctx.needs_input_grad[3] and not lazy, sync_group=ctx.sync_group, pre=pre,
                                                 absmax=slot)
        if slot is not None:
            _tag_absmax(dx, slot)               # (the convolution whose output x is reads it off its `gy`: same tensor object)
        if lazy:
            dres = _lazy_grad(("mask", dy, mask), x.shape, x.device, x.dtype)
        elif has_res and dres is None and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None


_NAN_PLACEHOLDER = {}


def _deferred_view(x: Tensor) -> Tensor:
    """Stands for relu(bn(x)) where that tensor is never written: NaN under every index (one element, zero strides), so that a
    reader that does not know the protocol cannot go unnoticed."""
    nan = _NAN_PLACEHOLDER.get((x.device, x.dtype))      # (its own element: never the address of one of `_lazy_grad`'s sentinels)
    if nan is None:
        nan = _NAN_PLACEHOLDER[(x.device, x.dtype)] = torch.full((1,), float("nan"), device=x.device, dtype=x.dtype)
    return nan.view(1, 1, 1, 1).expand(x.shape)


class _Materialize(torch.autograd.Function):
    """The output of a BatchNorm layer whose apply pass was left to its consumer, written after all (the reader turned out
    not to be the pass that computes it on the fly): peclr_bn2d_apply on the finished table."""

    @staticmethod
    def forward(ctx, placeholder, deferred, amax=None):
        x, ss, relu = deferred
        slot = _new_absmax(x) if amax is not None else None
        if slot is not None:
            amax[:] = [slot]
        return _capi.bn2d_apply(x, ss, relu=relu, absmax=slot)

    @staticmethod
    def backward(ctx, gy):
        return gy, None, None


def _materialized(x: Tensor, deferred) -> Tensor:
    amax = []
    y = _Materialize.apply(x, deferred, amax)
    if amax:
        _tag_absmax(y, amax[0])
    link = getattr(x, "_peclr_bn_link", None)
    if link:
        y._peclr_bn_link = link
    return y


class _BN2dReluPool(torch.autograd.Function):
    """Stem: BatchNorm2d -> ReLU -> MaxPool2d(3, 2, 1) without ever writing the un-pooled activation."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn: "FusedBatchNormAct2d", pre=None, amax=None):
        training = bn.training or not bn.track_running_stats
        sync = bn.sync_group if training else None
        rm, rv, nbt, shift = bn._stat_buffers(training)
        slot = _new_absmax(x) if amax is not None else None
        if slot is not None:
            amax[:] = [slot]
        y, x_at_max, code, save, ss = _capi.bn2d_pool_fwd(x, weight, bias, rm, rv, nbt, training, bn.eps,
                                                bn.momentum if bn.momentum is not None else 0.1, sync_group=sync, sync_shift=shift,
                                                pre=pre if training else None, absmax=slot)
        ctx.save_for_backward(x, x_at_max, code, save, ss)
        ctx.cfg = (training, sync)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x_at_max, code, save, ss = ctx.saved_tensors
        training, sync = ctx.cfg
        dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx, dgamma, dbeta = _capi.bn2d_pool_bwd(dy, x, x_at_max, code, save, ss, training, sync_group=sync)
        return dx, dgamma, dbeta, None, None, None


class _BN2dAddReluAvgPool(torch.autograd.Function):
    """Encoder tail (the last residual block's bn + identity + ReLU, then AdaptiveAvgPool2d((1, 1)) and
    flatten): returns the pooled fp32 [N, C] features; the [N, C, H, W] activation is never materialised,
    and the backward forms the pool's broadcast gradient on the fly."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, bn: "FusedBatchNormAct2d", pre=None):
        training = bn.training or not bn.track_running_stats
        sync = bn.sync_group if training else None
        rm, rv, nbt, shift = bn._stat_buffers(training)
        pooled, mask, save, ss = _capi.bn2d_avgpool_fwd(x, residual, weight, bias, rm, rv, nbt, training, bn.eps,
                                                        bn.momentum if bn.momentum is not None else 0.1, sync_group=sync,
                                                        sync_shift=shift, pre=pre if training else None)
        ctx.save_for_backward(x, mask, save, ss)
        ctx.cfg = (training, sync)
        return pooled

    @staticmethod
    def backward(ctx, d_pooled):
        x, mask, save, ss = ctx.saved_tensors
        training, sync = ctx.cfg
        slot = _new_absmax(x)
        dx, dgamma, dbeta, dres = _capi.bn2d_avgpool_bwd(d_pooled.float().contiguous(), x, mask, save, ss, training,
                                                         sync_group=sync, absmax=slot)
        if slot is not None:
            _tag_absmax(dx, slot)
        return dx, dgamma, dbeta, dres, None, None


# ---- weight gradients on a side stream.  In the backward pass a convolution's weight gradient (MFMA-bound) is
# needed only by the optimiser, while its input gradient feeds the next BatchNorm backward (HBM-bound) at once.
# With `enable_wgrad_overlap`, every convolution of the in-tree ResNet computes its weight gradient on a second
# HIP stream: it runs next to the glue kernels of the layers below it instead of in front of them, so the chip's
# MFMA pipes and its memory system are busy at the same time.  Such a gradient does NOT travel through autograd's
# AccumulateGrad (which would read it on the main stream): it is parked and handed to its parameter by
# `wgrad_join()`, which the caller runs after backward, once the main stream has waited for the side stream.
# Under hipGraph capture the side stream becomes a parallel branch of the graph.
class _WgradOverlap:
    stream = None        # torch.cuda.Stream or None (= mode off: everything on the current stream, through autograd)
    armed = False        # the mode only acts between `arm_wgrad_overlap()` (before a forward pass whose backward the
    #                      caller will follow with `wgrad_join()`) and that join: any other forward / backward in the
    #                      process -- a plain `loss.backward(); opt.step()` -- goes through autograd as usual
    parked: list = []    # (parameter, gradient computed on the side stream, tensors it read)


def enable_wgrad_overlap(enabled: bool = True):
    """Create (or drop) the side stream of the side-stream weight gradients; they act only while armed."""
    if enabled and _WgradOverlap.stream is None:
        _WgradOverlap.stream = torch.cuda.Stream()
    elif not enabled:
        wgrad_join()
        _WgradOverlap.stream = None


def arm_wgrad_overlap():
    """Before a forward pass whose backward will be followed by `wgrad_join()` (the Trainer does both)."""
    _WgradOverlap.armed = _WgradOverlap.stream is not None


def _overlap_stream():
    return _WgradOverlap.stream if _WgradOverlap.armed else None


def wgrad_join(on_joined=None):
    """After backward: wait for the side stream, then give every parked gradient to its parameter
    (`.grad = g`, or `.grad += g` when one is already there: accumulation windows, all-reduce bucket views), call
    `on_joined(parameter)` for each (the all-reduce buckets count their members off there: these gradients never pass
    AccumulateGrad, so no post-accumulate hook fires for them), and disarm."""
    _WgradOverlap.armed = False
    if not _WgradOverlap.parked:
        return
    torch.cuda.current_stream().wait_stream(_WgradOverlap.stream)
    with torch.no_grad():
        for param, g, _inputs in _WgradOverlap.parked:
            if param.grad is None:
                param.grad = g
            else:
                param.grad.add_(g)
            if on_joined is not None:
                on_joined(param)
    _WgradOverlap.parked.clear()


def _conv_wgrad(gy: Tensor, x: Tensor, weight: Tensor, stride, padding, param=None):
    """d(weight) of conv2d(x, weight) for output gradient gy.  Overlap off (or no parameter to park it for):
    computed here and returned.  Overlap on: issued on the side stream, parked for `param`, returns None."""
    def run():
        g = torch.ops.aten.convolution_backward(gy, x, weight.to(x.dtype), None, list(stride), list(padding), [1, 1], False,
                                                [0, 0], 1, [False, True, False])[1]
        return g.to((param if param is not None else weight).dtype)
    st = _overlap_stream()
    if st is None or param is None or not gy.is_cuda:
        return run()
    st.wait_stream(torch.cuda.current_stream())          # gy (and x) are ready where the side stream picks up
    with torch.cuda.stream(st):
        g = run()
        if g.stride() != param.stride():                 # the parameter's layout (channels_last weights)
            g = torch.empty_like(param).copy_(g)
    _WgradOverlap.parked.append((param, g, (gy, x)))     # inputs stay referenced until the join
    return None


class _Conv2dSplitBackward(torch.autograd.Function):
    """conv2d (no bias, groups 1, dilation 1) whose backward computes the input gradient on the current stream and
    the weight gradient through `_conv_wgrad` (parked on the side stream when overlap is on)."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, param):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (tuple(stride), tuple(padding), param)
        return F.conv2d(x, weight, None, stride, padding)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, padding, param = ctx.cfg
        gy = gy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dw = _conv_wgrad(gy, x, weight, stride, padding, param) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(gy, x, weight.to(x.dtype), None, list(stride), list(padding), [1, 1], False,
                                                     [0, 0], 1, [True, False, False])[0]
        if dw is not None and dw.dtype != weight.dtype:
            dw = dw.to(weight.dtype)
        return dx, dw, None, None, None


def _x6_pays(rows: int, n_out: int, k: int) -> bool:
    """Does peclr_gemm_x6_f32 beat MIOpen's fp32 1x1 convolution for a [rows, k] x [k, n_out] product?  Measured on
    ResNet-50's shapes (tools/exp/conv1x1_probe.py, and in the step): from K = 128 on when the 128-wide column tile is
    full and the 128 x 128 tiles fill the chip; the K = 64 and N = 64 shapes of layer1 are HBM-bound either way."""
    # (64-wide outputs -- layer1 -- have a 128 x 64-tile variant in the library that wins in isolation, 256 vs 345 us,
    # and loses 0.4 ms per step inside it, where MIOpen's kernels find their operands in the caches: not routed)
    if not ROUTING.gemm_x6:
        return False
    if (k >= ROUTING.x6_min_k or (ROUTING.force and k >= 64)) and k % 4 == 0 and n_out >= 128 and n_out % 4 == 0 and ROUTING.fills_chip(rows, n_out):
        return True
    # layer1 (64 <-> 256 channels, 8e5 rows): HBM-bound, the GEMM itself is a draw against MIOpen (290 vs 296 us, 269 vs 331,
    # input gradient 253 vs 322: tools/exp/layer1_probe.py) -- what pays is that the in-tree GEMM also delivers the next
    # BatchNorm's statistics / performs the previous one's backward reduction, each a pass over up to 822 MB
    return (ROUTING.x6_layer1 and ROUTING.gemm_x6p and ROUTING.streams_from_hbm(rows, k, n_out) and k >= 64 and k % 16 == 0
            and n_out >= 64 and n_out % 64 == 0)




def _x6_wgrad_pays(rows: int, cout: int, cin: int) -> bool:
    """peclr_gemm_x6t_f32 against MIOpen's fp32 1x1 weight gradient (tools/exp/conv1x1_probe.py, wgrad_probe.py): 150-205 us
    against 208-267 from layer2 on; the 64-wide gradients of layer1 (HBM-bound: 8e5 rows of 64 + 256 channels) on 64 x 256 /
    256 x 64 / 64 x 128 tiles."""
    wide = 128 if not (ROUTING.gemm_x6t and ROUTING.x6_layer1_wgrad) else 64
    return ROUTING.gemm_x6 and cout >= wide and cin >= wide and cout % 4 == 0 and cin % 4 == 0 and (rows >= 8192 or ROUTING.force)


def _wgrad_1x1_x6(gy: Tensor, x: Tensor, weight: Tensor, param=None):
    """d(weight) of a 1x1 / stride-1 convolution as dY^T X on the bf16 matrix cores (fp32 accuracy, deterministic
    split-K), shaped and strided like the weight; parked for `param` on the side stream when that mode is on."""
    n, cin, h, w = x.shape
    cout = gy.shape[1]

    def run():
        gy2, x2 = gy.permute(0, 2, 3, 1).reshape(n * h * w, cout), x.permute(0, 2, 3, 1).reshape(n * h * w, cin)
        dw = (_capi.gemm_x6t(gy2, x2, tag="conv1x1_wgrad") if ROUTING.gemm_x6t
              else _capi.gemm_x6_tn(gy2, x2, tag="conv1x1_wgrad"))
        ref = param if param is not None else weight
        return dw.as_strided(ref.shape, ref.stride())       # same memory, the parameter's (channels_last) strides

    st = _overlap_stream()
    if st is None or param is None:
        return run()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = run()
    _WgradOverlap.parked.append((param, g, (gy, x)))
    return None


def _stat_shift_for(bn, cout: int):
    """The shift a GEMM epilogue subtracts before it sums the statistics `bn` will finalize (None: `bn` cannot take
    statistics from its producer -- not a fused training-mode BatchNorm of that width, or no running mean to centre on).
    The running mean is the natural centre (the sums are exact in the shift; it only has to be near the batch mean, and
    identical on every rank under synchronised statistics); the re-run of a checkpointed block under synchronised
    statistics re-uses the copy its first run took."""
    if (not isinstance(bn, FusedBatchNormAct2d) or not bn.hip or not bn.affine or bn.num_features != cout
            or bn.running_mean is None or not bn.training or not bn.running_mean.is_cuda):
        return None
    # activation checkpointing: the re-run centres its sums where the first run did (the running mean has moved in between),
    # so that it reproduces the first run's statistics bit for bit -- 16-bit activations amplify a last-bit difference in a
    # scale into whole-ulp differences a few layers on; under synchronised statistics the same copy is the common shift
    kept = _ckpt_shift_of(bn)
    return kept if kept is not None else bn.running_mean


def _wgrad_3x3_x6(gy: Tensor, x: Tensor, weight: Tensor, param=None, stride: int = 1, taps: int = 9):
    """d(weight) of a 3x3 / padding-1 convolution (stride 1 or 2): nine dY^T X products with X read at the pixel each tap
    points at, all in one launch that splits dY once for the nine taps (peclr_gemm_x6t_f32, taps = 9; fp32 accuracy,
    fixed-order split-K: deterministic), written in the weight's own channels_last storage order [Cout][3][3][Cin].
    taps = 1 with stride 2: the 1x1 / stride-2 shortcut convolution (X read at every second pixel)."""
    n, cin, h, w = x.shape
    cout, ho, wo = gy.shape[1:]

    def run():
        gy2, x2 = gy.permute(0, 2, 3, 1).reshape(n * ho * wo, cout), x.permute(0, 2, 3, 1).reshape(n * h * w, cin)
        dw = _capi.gemm_x6t(gy2, x2, taps=taps, hw=(ho, wo), stride=stride, tag="conv3x3_wgrad" if taps == 9 else "conv1x1_wgrad")
        if taps == 9:
            return dw.view(cout, 3, 3, cin).permute(0, 3, 1, 2)      # = a channels_last [Cout, Cin, 3, 3] tensor
        ref = param if param is not None else weight
        return dw.as_strided(ref.shape, ref.stride())               # [Cout][Cin] in memory either way

    st = _overlap_stream()
    if st is None or param is None:
        return run()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = run()
    _WgradOverlap.parked.append((param, g, (gy, x)))
    return None


class _Conv1x1Gemm(torch.autograd.Function):
    """1x1 / stride-1 convolution of an NHWC fp32 tensor as the GEMM it is, on the bf16 matrix cores at fp32 accuracy:
    forward y[R, Cout] = x[R, Cin] . W^T and / or the input gradient dx[R, Cin] = dy[R, Cout] . W, each where `_x6_pays`
    (weight planes packed once per step when the convolution belongs to an X6PackGroup: peclr_gemm_x6p_f32; otherwise
    both operands split per workgroup: peclr_gemm_x6_f32); the other direction and small weight gradients stay on MIOpen."""

    @staticmethod
    def forward(ctx, x, weight, conv, use_fwd: bool, use_bwd: bool, stats=None, link=None, amax=None):
        """stats: None, or [bn] -- the BatchNorm2d that consumes the output; the GEMM epilogue then sums its statistics
        and the list comes back as [partial, n_split, shift, bn] (left untouched when that is not possible).
        amax: the device float holding max |x| (`_absmax_of`) -> the forward runs in pair arithmetic."""
        ctx.save_for_backward(x, weight)
        planes = _x6_planes(conv) if (use_fwd or use_bwd) else None
        ctx.cfg = (conv, use_bwd, planes)
        ctx.link = link
        if not use_fwd:
            return F.conv2d(x, weight)
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        x2 = x.permute(0, 2, 3, 1).reshape(n * h * w, cin)
        shift = _stat_shift_for(stats[0], cout) if (stats and planes is not None and ROUTING.bn_stats_in_gemm) else None
        if planes is not None:
            pl, kw = _pair_planes(conv, amax, 0, planes)
        if shift is not None:
            y, partial, ns = _capi.gemm_x6p(x2, pl, cout, tag="conv1x1_fwd", stat_shift=shift, **kw)
            stats[:] = [partial, ns, shift, stats[0]]
        elif planes is not None:
            y = _capi.gemm_x6p(x2, pl, cout, tag="conv1x1_fwd", **kw)
        else:
            y = _capi.gemm_x6(x2, weight.detach().reshape(cout, cin), tag="conv1x1_fwd")
        return y.view(n, h, w, cout).permute(0, 3, 1, 2)          # channels_last NCHW view of the NHWC result

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        conv, use_bwd, planes = ctx.cfg
        param = conv.weight if conv is not None else None
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        gy = gy.contiguous(memory_format=torch.channels_last)
        dw = None
        if ctx.needs_input_grad[1]:
            dw = (_wgrad_1x1_x6(gy, x, weight, param) if _x6_wgrad_pays(n * h * w, cout, cin)
                  else _conv_wgrad(gy, x, weight, (1, 1), (0, 0), param))
        dx = None
        if ctx.needs_input_grad[0]:
            if use_bwd:
                gy2 = gy.permute(0, 2, 3, 1).reshape(n * h * w, cout)
                link = ctx.link
                if planes is not None:
                    pl, kw = _pair_planes(conv, _absmax_of(gy), 1, planes)
                if planes is not None and link is not None and link[0].shape == x.shape and cin % 32 == 0:
                    dx, partial, ns = _capi.gemm_x6p(gy2, pl, cin, tag="conv1x1_dgrad", bn_bwd=link[:5], **kw)
                    _note_bn_bwd(dx, link, partial, ns)
                elif planes is not None:
                    dx = _capi.gemm_x6p(gy2, pl, cin, tag="conv1x1_dgrad", **kw)
                else:
                    wt = weight.detach().reshape(cout, cin).t().contiguous()          # [Cin][Cout]: K-contiguous B operand
                    dx = _capi.gemm_x6(gy2, wt, tag="conv1x1_dgrad")
                dx = dx.view(n, h, w, cin).permute(0, 3, 1, 2)
            else:
                dx = torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        return dx, dw, None, None, None, None, None, None




class _Conv3x3Gemm(torch.autograd.Function):
    """3x3 / stride-1 / padding-1 convolution of an NHWC fp32 tensor as an implicit GEMM on the bf16 matrix cores at
    fp32 accuracy (peclr_conv3x3_x6p_f32: rows = output pixels, K = 9 Cin in (tap, channel) order, the activation rows
    of a k-step read from the pixel its tap points at, the filter from planes packed once per step): forward and input
    gradient (the same kernel on the flipped filter); the weight gradient as nine dY^T X products in one launch
    (`_wgrad_3x3_x6`)."""

    @staticmethod
    def forward(ctx, x, weight, conv, stats=None, link=None, amax=None):
        ctx.save_for_backward(x, weight)
        planes = _x6_planes(conv)
        ctx.cfg = (conv, planes)
        ctx.link = link
        cout = weight.shape[0]
        shift = _stat_shift_for(stats[0], cout) if (stats and ROUTING.bn_stats_in_gemm) else None
        pl, kw = _pair_planes(conv, amax, 0, planes)
        tile_rows = ROUTING.conv3x3_pair_tile_rows if kw else ROUTING.conv3x3_tile_rows
        if shift is not None:
            y, partial, ns = _capi.conv3x3_x6p(x, pl, cout, tag="conv3x3_fwd", tile_rows=tile_rows, stat_shift=shift, **kw)
            stats[:] = [partial, ns, shift, stats[0]]
            return y
        return _capi.conv3x3_x6p(x, pl, cout, tag="conv3x3_fwd", tile_rows=tile_rows, **kw)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        conv, planes = ctx.cfg
        gy = gy.contiguous(memory_format=torch.channels_last)
        dw = None
        if ctx.needs_input_grad[1]:
            in_tree = (ROUTING.gemm_x6t and ROUTING.conv3x3_wgrad_x6 and weight.is_contiguous(memory_format=torch.channels_last)
                       and x.shape[3] >= 6 and x.shape[1] % 4 == 0 and gy.shape[1] % 4 == 0
                       and gy.shape[1] >= (64 if ROUTING.x6_layer1_wgrad else 128))   # (64 output channels: the 64 x 64 block with two tap halves)
            dw = _wgrad_3x3_x6(gy, x, weight, conv.weight) if in_tree else _conv_wgrad(gy, x, weight, (1, 1), (1, 1), conv.weight)
        dx = None
        if ctx.needs_input_grad[0]:
            link = ctx.link
            pl, kw = _pair_planes(conv, _absmax_of(gy), 1, planes)
            tile_rows = ROUTING.conv3x3_pair_tile_rows if kw else ROUTING.conv3x3_tile_rows
            if link is not None and link[0].shape == x.shape and x.shape[1] % 32 == 0:
                # dx is the gradient arriving at the BatchNorm layer whose output x is: reduce it in the epilogue
                dx, partial, ns = _capi.conv3x3_x6p(gy, pl, x.shape[1], flip=True, tag="conv3x3_dgrad", tile_rows=tile_rows,
                                                    bn_bwd=link[:5], **kw)
                _note_bn_bwd(dx, link, partial, ns)
            else:
                dx = _capi.conv3x3_x6p(gy, pl, x.shape[1], flip=True, tag="conv3x3_dgrad", tile_rows=tile_rows, **kw)
        return dx, dw, None, None, None, None




class _ConvS2Gemm(torch.autograd.Function):
    """Stride-2 3x3 (padding 1) / 1x1 convolution of an NHWC fp32 tensor: forward on the six-product kernel
    (peclr_conv_s2_x6p_f32: the k-step's tap and the stride select the source pixel; BatchNorm statistics of the output
    in the epilogue), weight gradient on peclr_gemm_x6t_f32 with stride 2.  Input gradient: 3x3 -- one dense implicit GEMM
    per parity class of input pixels (peclr_conv3x3_s2_dgrad_x6p_f32); 1x1 shortcut -- kept compact for the block's
    entry-gradient GEMM (`_compact_grad`) where that is its consumer, else MIOpen."""

    @staticmethod
    def forward(ctx, x, weight, conv, stats=None, compact=False, link=None, amax=None):
        ctx.save_for_backward(x, weight)
        ctx.conv = conv
        ctx.compact = compact
        ctx.link = link
        planes = _x6_planes(conv)
        cout, taps = weight.shape[0], weight.shape[2] * weight.shape[3]
        shift = _stat_shift_for(stats[0], cout) if (stats and ROUTING.bn_stats_in_gemm) else None
        pl, kw = _pair_planes(conv, amax, 0, planes)
        if shift is not None:
            y, partial, ns = _capi.conv_s2_x6p(x, pl, cout, taps, tag="conv_s2_fwd", stat_shift=shift, tile_rows=ROUTING.s2_tile_rows if taps == 9 else 0, **kw)
            stats[:] = [partial, ns, shift, stats[0]]
            return y
        return _capi.conv_s2_x6p(x, pl, cout, taps, tag="conv_s2_fwd", tile_rows=ROUTING.s2_tile_rows if taps == 9 else 0, **kw)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        conv = ctx.conv
        gy = gy.contiguous(memory_format=torch.channels_last)
        pad = list(conv.padding)
        dw = None
        if ctx.needs_input_grad[1]:
            taps = weight.shape[2] * weight.shape[3]
            in_tree = (ROUTING.gemm_x6t and ROUTING.conv_s2_wgrad_x6 and weight.is_contiguous(memory_format=torch.channels_last)
                       and x.shape[2] == 2 * gy.shape[2] and x.shape[3] == 2 * gy.shape[3] and gy.shape[3] >= 6
                       and x.shape[1] % 4 == 0 and gy.shape[1] % 4 == 0 and gy.shape[1] >= 64 and x.shape[1] >= 64)
            dw = (_wgrad_3x3_x6(gy, x, weight, conv.weight, stride=2, taps=taps) if in_tree
                  else _conv_wgrad(gy, x, weight, (2, 2), pad, conv.weight))
        dx = None
        if ctx.needs_input_grad[0]:
            if weight.shape[2] == 1 and x.shape[2] == 2 * gy.shape[2] and x.shape[3] == 2 * gy.shape[3]:
                # 1x1 shortcut: dY . W over the OUTPUT pixels only (three quarters of the input gradient are zeros).  Where it
                # goes straight into the block's fused entry gradient it stays compact (the consumer adds it at the even
                # pixels: no 4x larger tensor); elsewhere (BasicBlock networks: the block input is a plain tensor) it is
                # scattered into zeros here -- never MIOpen's input gradient, whose fp32 1x1 / stride-2 solver adds with float
                # atomics (a different last bit every run: tools/exp/two_outcome.py)
                n, cout, ho, wo = gy.shape
                pl, kw = _pair_planes(conv, _absmax_of(gy), 1, _x6_planes(conv))
                dc = _capi.gemm_x6p(gy.permute(0, 2, 3, 1).reshape(n * ho * wo, cout), pl, x.shape[1], tag="conv_s2_dgrad", **kw)
                dx = _compact_grad(dc, x.shape) if (ctx.compact and not torch.is_anomaly_enabled()) else _expand_compact(dc, x.shape)
            elif (ROUTING.conv_s2_dgrad_x6 and weight.shape[2] == 3 and x.shape[2] == 2 * gy.shape[2] and x.shape[3] == 2 * gy.shape[3]
                  and x.shape[1] % 64 == 0 and gy.shape[1] % 16 == 0):
                # 3x3: one dense implicit GEMM per parity class of input pixels (1, 2, 2, 4 taps); dx is the gradient arriving
                # at the BatchNorm layer whose output x is: reduced in the epilogue
                pl, kw = _pair_planes(conv, _absmax_of(gy), 1, _x6_planes(conv))
                link = ctx.link
                if link is not None and link[0].shape == x.shape and x.shape[1] % 32 == 0:
                    dx, partial, ns = _capi.conv3x3_s2_dgrad_x6p(gy, pl, x.shape[1], bn_bwd=link[:5], tile_rows=ROUTING.s2_tile_rows, **kw)
                    _note_bn_bwd(dx, link, partial, ns)
                else:
                    dx = _capi.conv3x3_s2_dgrad_x6p(gy, pl, x.shape[1], tile_rows=ROUTING.s2_tile_rows, **kw)
            else:
                dx = torch.ops.aten.convolution_backward(gy, x, weight, None, [2, 2], pad, [1, 1], False, [0, 0], 1, [True, False, False])[0]
        return dx, dw, None, None, None, None, None


# ---- gradients handed to a block's entry-gradient GEMM in another form than a dense tensor.  An autograd function may only
# return a tensor of its input's shape.  Two producers would write such a tensor only for `_ForkConv1x1.backward` to read it
# back as its addend: the 1x1 / stride-2 shortcut of a layer's first block (three quarters of its input gradient are zeros:
# the compact [N, H/2, W/2, C] product is all there is to it) and the last BatchNorm of a block with an identity shortcut
# (the residual's gradient is relu'(out) * d(out): the GEMM can read d(out) and the 1-bit mask itself).  When their input is
# the identity output of `_ForkConv1x1` (whose backward is the only consumer of that gradient, and understands the
# protocol), they return a zero-stride NaN view of the right shape instead -- no memory, and loudly wrong should anything
# else ever consume or accumulate it -- and park the real payload here under the view's address.
_COMPACT = {}
_NAN_RING = {}


def _lazy_grad(payload, shape, device, dtype=torch.float32) -> Tensor:
    ring = _NAN_RING.get((device, dtype))        # (autograd casts a gradient of another dtype: that would materialise the view)
    if ring is None:
        ring = _NAN_RING[(device, dtype)] = [torch.full((64,), float("nan"), device=device, dtype=dtype), 0]
    buf, at = ring
    ring[1] = (at + 1) % 64
    sentinel = buf[at:at + 1].view(1, 1, 1, 1).expand(shape)
    _COMPACT[sentinel.data_ptr()] = (payload, tuple(shape))
    _drain_after_backward()
    return sentinel


def _compact_grad(dc: Tensor, shape) -> Tensor:
    return _lazy_grad(("s2", dc), shape, dc.device, dc.dtype)


def _take_lazy(g: Tensor):
    """The parked payload -- ("s2", compact gradient) or ("mask", dy, bit mask) -- if `g` is one of `_lazy_grad`'s views."""
    if g is None or g.dim() != 4 or any(g.stride()) or not _COMPACT:
        return None
    hit = _COMPACT.get(g.data_ptr())
    if hit is None or hit[1] != tuple(g.shape):
        return None
    del _COMPACT[g.data_ptr()]
    return hit[0]


def _take_compact(g: Tensor):
    hit = _take_lazy(g)
    return None if hit is None else _dense_of(hit, g.shape) if hit[0] != "s2" else hit[1]


def _dense_of(payload, shape) -> Tensor:
    """The dense gradient a payload stands for (fallback paths)."""
    if payload[0] == "s2":
        return _expand_compact(payload[1], shape)
    _, dy, mask = payload
    n, c, h, w = shape
    bits = (mask.view(n * h * w, c // 32, 1) >> torch.arange(32, device=mask.device, dtype=torch.int32)) & 1
    d = dy.permute(0, 2, 3, 1).reshape(n * h * w, c) * bits.view(n * h * w, c).to(dy.dtype)
    return d.view(n, h, w, c).permute(0, 3, 1, 2)


def _expand_compact(dc: Tensor, shape) -> Tensor:
    """The dense form (fallback paths): zeros with the compact gradient at the even pixels."""
    n, c, h, w = shape
    full = torch.zeros(shape, device=dc.device, dtype=dc.dtype).contiguous(memory_format=torch.channels_last)
    full[:, :, ::2, ::2] = dc.view(n, h // 2, w // 2, c).permute(0, 3, 1, 2)
    return full


def _bn_link_of(x: Tensor):
    """(x_bn, save, scale_shift, mask, relu, token) if `x` is the output of a fused BatchNorm layer whose backward reduction
    a consumer's input-gradient GEMM may perform, else None."""
    link = getattr(x, "_peclr_bn_link", None)
    return tuple(link) if link and ROUTING.bn_bwd_in_gemm else None


_DRAIN_QUEUED = False
_NESTED_BACKWARD = 0       # > 0 inside the backward pass a checkpointed block runs over its re-run graph


def _drain_after_backward():
    """Called by the producers of side-channel entries (they run inside a backward pass): have the autograd engine call
    `end_backward()` when THIS backward pass is over, so that any training loop -- not only `Trainer`, which also calls it --
    leaves no stale entry pinning device tensors behind (should a pass end without running its callbacks, the token / version
    checks still keep a stale entry from ever being used, and the next pass drains it)."""
    global _DRAIN_QUEUED
    if _DRAIN_QUEUED or _NESTED_BACKWARD:
        return
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_drained)
        _DRAIN_QUEUED = True          # (reset by the callback -- or by end_backward(), should a pass raise and skip its callbacks)
    except RuntimeError:        # not inside a backward pass (a kernel-level test calling a backward function by hand)
        pass


last_backward_leftovers = 0     # entries the engine-queued drain of the most recent backward pass had to drop (0 for the in-tree ResNets)


def _drained():
    global _DRAIN_QUEUED, last_backward_leftovers
    _DRAIN_QUEUED = False
    last_backward_leftovers = end_backward()


def _note_bn_bwd(dx: Tensor, link, partial, ns):
    _BN_BWD_STATS[dx.data_ptr()] = (link[5], partial, ns, dx._version)
    _drain_after_backward()


def end_backward(strict: bool = False) -> int:
    """After every backward pass (the Trainer calls it): drop what the side channels still hold -- a BatchNorm reduction
    whose gradient tensor was merged into another consumer's buffer and never arrived under its own address, a lazy
    payload whose NaN view nobody took -- so that nothing pins device memory or meets a recycled address later.  Returns
    the number of entries dropped; strict=True raises instead (tests: the in-tree ResNets leave none behind)."""
    global _DRAIN_QUEUED
    _DRAIN_QUEUED = False        # (a pass that raised skipped its engine callbacks: the next one queues its drain again)
    n = len(_BN_BWD_STATS) + len(_COMPACT)
    _BN_BWD_STATS.clear()
    _COMPACT.clear()
    if n and strict:
        raise _capi.PeclrHipError(f"{n} gradient hand-overs were never consumed")
    return n


def _attach_stats(y: Tensor, stats):
    """Hand the statistics a GEMM epilogue summed to the BatchNorm that consumes `y` (FusedBatchNormAct2d.forward looks for them)."""
    if stats is not None and len(stats) == 4:
        y._peclr_bn_stats = tuple(stats)
    return y


def _wgrad_h(gy: Tensor, x: Tensor, conv, stride: int):
    """d(weight) (fp32, the parameter's layout) of a 16-bit convolution.  In-tree (peclr_wgrad_h: fixed-order slabs, fp32
    out straight into the master weight's layout) where the kernel takes the shape, else MIOpen's 16-bit weight gradient
    (+ its cast to fp32)."""
    w = conv.weight
    taps = w.shape[2] * w.shape[3]
    fn = getattr(_capi, "wgrad_h", None)
    if (fn is not None and ROUTING.wgrad16 and w.is_contiguous(memory_format=torch.channels_last)
            and _capi.wgrad_h_ok(gy, x, taps, stride)):
        def run():
            dw = fn(gy, x, taps, stride)                  # [Cout, taps * Cin] fp32
            if taps == 9:
                return dw.view(w.shape[0], 3, 3, w.shape[1]).permute(0, 3, 1, 2)    # = a channels_last [Cout, Cin, 3, 3] tensor
            return dw.as_strided(w.shape, w.stride())
        st = _overlap_stream()
        if st is None:
            return run()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            g = run()
        _WgradOverlap.parked.append((w, g, (gy, x)))
        return None
    return _conv_wgrad(gy, x, w, conv.stride, conv.padding, w)


class _ConvH(torch.autograd.Function):
    """A residual block's convolution on 16-bit (bf16 / fp16 autocast) NHWC activations, in-tree (csrc/conv_h.hip): 1x1 or
    3x3 / padding 1, stride 1 or 2 -- forward with the consuming BatchNorm's statistics in the epilogue, input gradient with
    the producing BatchNorm's backward reduction in the epilogue, both from weight planes packed once per step out of the
    fp32 master weights.  Same hand-over protocols as the fp32 classes above (`stats`, `link`, compact shortcut gradient)."""

    @staticmethod
    def forward(ctx, x, weight, conv, stats=None, link=None, compact=False):
        ctx.save_for_backward(x)
        ctx.conv, ctx.link, ctx.compact = conv, link, compact
        planes = conv.x6_group.planes(conv, x.dtype)
        ctx.planes = planes
        cout, taps, stride = conv.out_channels, conv.kernel_size[0] * conv.kernel_size[1], conv.stride[0]
        shift = _stat_shift_for(stats[0], cout) if (stats and ROUTING.bn_stats_in_gemm) else None
        n, cin, h, w = x.shape
        if taps == 1 and stride == 1:
            x2 = x.permute(0, 2, 3, 1).reshape(n * h * w, cin)
            if shift is not None:
                y, partial, ns = _capi.gemm_h(x2, planes[0], cout, tag="conv1x1_fwd", stat_shift=shift)
                stats[:] = [partial, ns, shift, stats[0]]
            else:
                y = _capi.gemm_h(x2, planes[0], cout, tag="conv1x1_fwd")
            return y.view(n, h, w, cout).permute(0, 3, 1, 2)
        tag = "conv3x3_fwd" if stride == 1 else "conv_s2_fwd"
        if shift is not None:
            y, partial, ns = _capi.conv_h(x, planes[0], cout, taps=taps, stride=stride, tag=tag, stat_shift=shift)
            stats[:] = [partial, ns, shift, stats[0]]
            return y
        return _capi.conv_h(x, planes[0], cout, taps=taps, stride=stride, tag=tag)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        conv, planes, link = ctx.conv, ctx.planes, ctx.link
        gy = gy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        n, cin, h, w = x.shape
        cout, taps, stride = conv.out_channels, conv.kernel_size[0] * conv.kernel_size[1], conv.stride[0]
        dw = _wgrad_h(gy, x, conv, stride) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            fuse = link[:5] if (link is not None and link[0].shape == x.shape and link[0].dtype == x.dtype and cin % 32 == 0) else None
            if taps == 1 and stride == 1:
                gy2 = gy.permute(0, 2, 3, 1).reshape(n * h * w, cout)
                out = _capi.gemm_h(gy2, planes[1], cin, tag="conv1x1_dgrad", bn_bwd=fuse)
            elif taps == 9 and stride == 1:
                out = _capi.conv_h(gy, planes[1], cin, flip=True, tag="conv3x3_dgrad", bn_bwd=fuse)
            elif taps == 9:
                out = _capi.conv3x3_s2_dgrad_h(gy, planes[1], cin, bn_bwd=fuse)
            elif h == 2 * gy.shape[2] and w == 2 * gy.shape[3]:
                # 1x1 / stride-2 shortcut: dY . W over the OUTPUT pixels only; the block's entry-gradient GEMM adds it at the even
                # pixels, any other consumer gets it scattered into zeros (deterministic, unlike MIOpen's atomics)
                ho, wo = gy.shape[2:]
                dc = _capi.gemm_h(gy.permute(0, 2, 3, 1).reshape(n * ho * wo, cout), planes[1], cin, tag="conv_s2_dgrad")
                lazy = ctx.compact and not torch.is_anomaly_enabled()
                return (_compact_grad(dc, x.shape) if lazy else _expand_compact(dc, x.shape)), dw, None, None, None, None
            else:
                w16 = conv.weight.detach().to(x.dtype)
                return (torch.ops.aten.convolution_backward(gy, x, w16, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0],
                        dw, None, None, None, None)
            if fuse is not None:
                dx, partial, ns = out
                _note_bn_bwd(dx, link, partial, ns)
            else:
                dx = out
            if taps == 1:
                dx = dx.view(n, h, w, cin).permute(0, 3, 1, 2)
        return dx, dw, None, None, None, None


def _stem_wgrad(gy: Tensor, x: Tensor, param):
    """d(weight) of the stem (peclr_stem_wgrad: fixed-order slabs, fp32 gradient of the fp32 master weight) in the parameter's
    memory format; parked for `param` on the side stream when that mode is on."""
    def run():
        dw = _capi.stem_wgrad(gy, x)
        return torch.empty_like(param).copy_(dw)                   # (9 408 elements: into the parameter's own strides)

    st = _overlap_stream()
    if st is None:
        return run()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = run()
    _WgradOverlap.parked.append((param, g, (gy, x)))
    return None


class _StemConv(torch.autograd.Function):
    """The encoder's 7x7 / stride-2 / padding-3 stem on fp32 NHWC images, in-tree (peclr_stem_conv7x7_s2): fp32 output at fp32
    accuracy, or -- under bf16 / fp16 autocast -- 16-bit output from the operands rounded inside the kernel (no cast pass over
    the images); the statistics of the BatchNorm behind it in the epilogue.  Weight gradient: peclr_stem_wgrad (the images need no
    gradient; one that is asked for comes from MIOpen)."""

    @staticmethod
    def forward(ctx, x, weight, conv, stats=None):
        half = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None
        dtype = half if half in _HALF else torch.float32
        planes = conv._stem_planes(dtype)
        ctx.save_for_backward(x, weight)
        ctx.conv, ctx.dtype = conv, dtype
        shift = _stat_shift_for(stats[0], 64) if (stats and ROUTING.bn_stats_in_gemm) else None
        if shift is not None:
            y, partial, ns = _capi.stem_conv(x, planes, stat_shift=shift)
            stats[:] = [partial, ns, shift, stats[0]]
            return y
        return _capi.stem_conv(x, planes)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        conv = ctx.conv
        gy = gy.to(ctx.dtype).contiguous(memory_format=torch.channels_last)
        dw = None
        if ctx.needs_input_grad[1]:
            if ROUTING.stem_wgrad:
                dw = _stem_wgrad(gy, x, conv.weight)
            else:
                xin = x if ctx.dtype == torch.float32 else x.to(ctx.dtype)          # (what autocast's convolution saw)
                dw = _conv_wgrad(gy, xin, weight, (2, 2), (3, 3), conv.weight)
        dx = None
        if ctx.needs_input_grad[0]:
            xin = x if ctx.dtype == torch.float32 else x.to(ctx.dtype)
            dx = torch.ops.aten.convolution_backward(gy, xin, weight.to(ctx.dtype), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0].to(x.dtype)
        if dw is not None and dw.dtype != weight.dtype:
            dw = dw.to(weight.dtype)
        return dx, dw, None, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict) that routes through `_Conv2dSplitBackward` while the side-stream
    weight gradients are enabled and the input is a channels_last HIP tensor; the stock op otherwise."""

    hip_gemm = False   # enable_hip_batchnorm: fp32 1x1 / stride-1 convolutions as GEMMs on the bf16 matrix cores
    hip_stem = False   # enable_hip_batchnorm: this is the 7x7 / stride-2 stem of 3-channel images (csrc/stem.hip)

    def _stem_planes(self, dtype):
        """The stem filter packed for `dtype` (fp32: three bf16 planes; bf16 / fp16: one plane), fresh: re-packed when the weight
        changed since the last pack (`_version`, fused optimiser epoch, storage) and once per hipGraph capture -- the protocol
        of `X6PackGroup.planes`."""
        sets = self.__dict__.setdefault("_stem_sets", {})
        st = sets.get(dtype)
        w = self.weight
        key = (w.data_ptr(), w._version, _capi.WEIGHTS_EPOCH)
        cap = _capi.capture_id()
        if st is None or st[0].weight.data_ptr() != w.data_ptr():
            if cap != 0:
                raise _capi.PeclrHipError("stem: the filter planes of this precision do not exist yet and cannot be created while a "
                                          "hipGraph is being captured -- run one eager forward pass (same autocast dtype) first")
            st = sets[dtype] = [_capi.StemPlanes(w.detach(), dtype), None, 0]
        stale = st[1] != key
        if cap != st[2]:
            stale = stale or cap != 0
            st[2] = cap
        if stale:
            st[0].pack()
            st[1] = key
        return st[0]

    def forward(self, x: Tensor, stats_for=None, sole_consumer: bool = False) -> Tensor:
        """stats_for: the BatchNorm2d that consumes the output -- when this convolution runs as an in-tree GEMM its
        epilogue sums that layer's training statistics (one pass over the activation less).
        sole_consumer: the caller guarantees that NOTHING else reads `x` (bn -> conv inside a residual block): only then
        is this convolution's input gradient THE gradient arriving at the BatchNorm layer that produced x, and only then
        may the input-gradient GEMM perform that layer's backward reduction in its epilogue.  A block input (x also feeds
        the shortcut) is not: BasicBlock.conv1 passes False."""
        bn_link = _bn_link_of if sole_consumer else (lambda t: None)
        deferred = getattr(x, "_peclr_deferred", None)
        if deferred is not None:
            x = _materialized(x, tuple(deferred))       # a BatchNorm layer's placeholder (its apply pass was left to another reader): written after all
        if (self.hip_stem and ROUTING.stem and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3
                and x.shape[2] >= 8 and x.shape[3] >= 8 and x.is_contiguous(memory_format=torch.channels_last)
                and self.weight.is_cuda and self.weight.dtype == torch.float32):
            stats = [stats_for] if stats_for is not None else None
            return _attach_stats(_StemConv.apply(x, self.weight, self, stats), stats)
        if self.hip_gemm and _h_ok(self, x) and (x.shape[2] % self.stride[0] == 0 and x.shape[3] % self.stride[1] == 0):
            stats = [stats_for] if stats_for is not None else None
            grad = torch.is_grad_enabled() and x.requires_grad
            compact = (ROUTING.s2_dgrad_compact and self.kernel_size == (1, 1) and self.stride == (2, 2)
                       and getattr(x, "_peclr_compact_ok", False) and grad)
            return _attach_stats(_ConvH.apply(x, self.weight, self, stats, bn_link(x) if grad else None, compact), stats)
        if (self.hip_gemm and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")
                and self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0) and self.groups == 1
                and self.bias is None and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
            rows = x.shape[0] * x.shape[2] * x.shape[3]
            use_fwd = _x6_pays(rows, self.out_channels, self.in_channels)
            use_bwd = _x6_pays(rows, self.in_channels, self.out_channels) and torch.is_grad_enabled() and x.requires_grad
            use_wgrad = (_x6_wgrad_pays(rows, self.out_channels, self.in_channels) and torch.is_grad_enabled()
                         and self.weight.requires_grad)
            if use_fwd or use_bwd or use_wgrad:
                stats = [stats_for] if (stats_for is not None and use_fwd) else None
                return _attach_stats(_Conv1x1Gemm.apply(x, self.weight, self, use_fwd, use_bwd, stats, bn_link(x) if use_bwd else None, _absmax_of(x)), stats)
        if (self.hip_gemm and ROUTING.conv_s2_x6 and ROUTING.gemm_x6p and getattr(self, "x6_group", None) is not None and self.stride == (2, 2)
                and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda") and x.dim() == 4
                and x.is_contiguous(memory_format=torch.channels_last) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                and (x.shape[0] * x.shape[2] * x.shape[3] >= 32768 or ROUTING.force)):
            stats = [stats_for] if stats_for is not None else None
            compact = (ROUTING.s2_dgrad_compact and self.kernel_size == (1, 1) and getattr(x, "_peclr_compact_ok", False)
                       and torch.is_grad_enabled() and x.requires_grad)
            link = bn_link(x) if (torch.is_grad_enabled() and x.requires_grad and self.kernel_size == (3, 3)) else None
            return _attach_stats(_ConvS2Gemm.apply(x, self.weight, self, stats, compact, link, _absmax_of(x)), stats)
        if (self.hip_gemm and ROUTING.conv3x3_x6 and ROUTING.gemm_x6p and getattr(self, "x6_group", None) is not None and self.kernel_size == (3, 3)
                and self.stride == (1, 1)
                and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda") and x.dim() == 4
                and x.is_contiguous(memory_format=torch.channels_last)
                and (x.shape[0] * x.shape[2] * x.shape[3] >= 8192 or ROUTING.force)):
            stats = [stats_for] if stats_for is not None else None
            return _attach_stats(_Conv3x3Gemm.apply(x, self.weight, self, stats, bn_link(x), _absmax_of(x)), stats)
        if (_overlap_stream() is not None and x.is_cuda and torch.is_grad_enabled() and self.bias is None
                and self.groups == 1 and self.dilation == (1, 1) and isinstance(self.padding, tuple)
                and x.is_contiguous(memory_format=torch.channels_last) and self.weight.requires_grad):
            w = self.weight
            if torch.is_autocast_enabled("cuda"):
                half = torch.get_autocast_dtype("cuda")
                x, w = x.to(half), w.to(half)            # what autocast does for conv2d
            return _Conv2dSplitBackward.apply(x, w, self.stride, self.padding, self.weight)
        return super().forward(x)


def _entry_gemm(a: Tensor, planes: Tensor, n: int, addend=None, **kw):
    """The block's entry gradient a . W (+ the shortcut's gradient, + the BatchNorm backward reduction): peclr_gemm_x6p_f32.
    (Round 5 tried a barrier-free streaming kernel for the HBM-bound shapes here -- tools/exp/gemm_x6s.hip, bit-identical
    output -- and measured it slower: docs/history.md.)"""
    return _capi.gemm_x6p(a, planes, n, addend, tag="conv1x1_dgrad_add_x6", **kw)


class _ForkConv1x1(torch.autograd.Function):
    """Bottleneck entry: (x, W) -> (conv1x1(x, W), x).  The block input feeds both the first convolution and
    the identity branch, so its gradient is dY W + d_identity: autograd runs MIOpen's dgrad and then an
    elementwise add over the block input (3.6 % of the fp32 step); here both are ONE fp32 GEMM with the
    identity gradient added in the epilogue.  Small shapes' forward and weight gradient stay on MIOpen."""

    @staticmethod
    def forward(ctx, x, weight, conv, stats=None, link=None, flags=None, amax=None):
        ctx.save_for_backward(x, weight)
        ctx.param = weight if isinstance(weight, nn.Parameter) else None
        ctx.link = link
        ctx.conv = conv
        n, cin, h, w = x.shape
        cmid = weight.shape[0]
        r = n * h * w
        use_fwd = x.dtype == torch.float32 and _x6_pays(r, cmid, cin)
        # (two rounds of the chip's CUs: below that MIOpen's dgrad + the elementwise add are not slower)
        use_bwd = (x.dtype == torch.float32 and ROUTING.gemm_x6 and cin >= 128 and ROUTING.fills_chip(r, cin, rounds=2.0)
                   and (cmid >= ROUTING.x6_min_k or (ROUTING.x6_layer1_fork and ROUTING.gemm_x6p and ROUTING.streams_from_hbm(r, cmid, cin)
                                                     and cmid >= 64 and cmid % 16 == 0 and cin % 128 == 0)))
        ctx.planes = _x6_planes(conv) if (use_fwd or use_bwd) else None
        ctx.use_bwd = use_bwd
        if flags is not None:       # tells fork_conv1x1 whether the backward takes compact shortcut gradients (the x6p GEMM)
            flags.append(bool(use_bwd and ctx.planes is not None))
        if use_fwd:
            x2 = x.permute(0, 2, 3, 1).reshape(r, cin)
            shift = _stat_shift_for(stats[0], cmid) if (stats and ctx.planes is not None and ROUTING.bn_stats_in_gemm) else None
            if ctx.planes is not None:
                pl, kw = _pair_planes(conv, amax, 0, ctx.planes)
            if shift is not None:
                y, partial, ns = _capi.gemm_x6p(x2, pl, cmid, tag="conv1x1_fwd", stat_shift=shift, **kw)
                stats[:] = [partial, ns, shift, stats[0]]
            elif ctx.planes is not None:
                y = _capi.gemm_x6p(x2, pl, cmid, tag="conv1x1_fwd", **kw)
            else:
                y = _capi.gemm_x6(x2, weight.detach().reshape(cmid, cin), tag="conv1x1_fwd")
            return y.view(n, h, w, cmid).permute(0, 3, 1, 2), x.view_as(x)
        return F.conv2d(x, weight), x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gid):
        x, weight = ctx.saved_tensors
        n, cin, h, w = x.shape
        cmid = weight.shape[0]
        dw = None
        gy = gy.to(x.dtype)
        if ctx.needs_input_grad[1]:
            gyc = gy.contiguous(memory_format=torch.channels_last)
            dw = (_wgrad_1x1_x6(gyc, x, weight, ctx.param) if (x.dtype == torch.float32 and _x6_wgrad_pays(n * h * w, cmid, cin))
                  else _conv_wgrad(gyc, x, weight, (1, 1), (0, 0), ctx.param))
        dx = None
        if ctx.needs_input_grad[0]:
            gy = gy.contiguous(memory_format=torch.channels_last)
            r = n * h * w
            a = gy.permute(0, 2, 3, 1).reshape(r, cmid)           # NHWC storage seen as [R, Cmid]: a view
            lazy = _take_lazy(gid)          # the shortcut's gradient in its compact / (dy, mask) form (see `_lazy_grad`)
            if lazy is not None and ctx.use_bwd and ctx.planes is not None and x.dtype == torch.float32:
                if lazy[0] == "s2":
                    kw = dict(addend=lazy[1], addend_s2=(h, w))
                else:
                    kw = dict(addend=lazy[1].contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(r, cin), addend_mask=lazy[2])
                link = ctx.link
                pl, pkw = _pair_planes(ctx.conv, _absmax_of(gy), 1, ctx.planes)
                kw.update(pkw)
                if link is not None and link[0].shape == x.shape and cin % 32 == 0:
                    out, partial, ns = _entry_gemm(a, pl, cin, bn_bwd=link[:5], **kw)
                    _note_bn_bwd(out, link, partial, ns)
                else:
                    out = _entry_gemm(a, pl, cin, **kw)
                return out.view(n, h, w, cin).permute(0, 3, 1, 2), dw, None, None, None, None, None
            if lazy is not None:
                gid = _dense_of(lazy, x.shape)
            gid = gid.to(x.dtype).contiguous(memory_format=torch.channels_last)
            d = gid.permute(0, 2, 3, 1).reshape(r, cin)
            if x.dtype in (torch.bfloat16, torch.float16):       # autocast backbone: 16-bit MFMA, fp32 accumulate
                wt = torch.empty((cin, cmid), device=x.device, dtype=x.dtype)
                wt.copy_(weight.detach().reshape(cmid, cin).t())      # transpose + cast in ONE launch
                out = _capi.gemm_add_half(a, wt, d, tag="conv1x1_dgrad_add")
            elif ctx.use_bwd and ctx.planes is not None:
                # fp32 on the bf16 matrix cores (exact 3-way split, six products), weight planes packed once per step;
                # the result is the gradient arriving at the previous block's last BatchNorm: reduced in the epilogue
                link = ctx.link
                pl, pkw = _pair_planes(ctx.conv, _absmax_of(gy), 1, ctx.planes)
                if link is not None and link[0].shape == x.shape and cin % 32 == 0:
                    out, partial, ns = _entry_gemm(a, pl, cin, d, bn_bwd=link[:5], **pkw)
                    _note_bn_bwd(out, link, partial, ns)
                else:
                    out = _entry_gemm(a, pl, cin, d, **pkw)
            elif ctx.use_bwd:
                wt = weight.detach().reshape(cmid, cin).t().contiguous()
                out = _capi.gemm_x6(a, wt, d, tag="conv1x1_dgrad_add_x6")
            else:
                out = _capi.gemm_add(_capi.GEMM_NN, a, weight.reshape(cmid, cin), d, tag="conv1x1_dgrad_add")
            dx = out.view(n, h, w, cin).permute(0, 3, 1, 2)       # back to a channels_last NCHW tensor
        return dx, dw, None, None, None, None, None


class _ForkConvH(torch.autograd.Function):
    """Bottleneck entry on 16-bit activations, in-tree: (x, W) -> (conv1x1(x, W), x); forward with bn1's statistics, backward
    dY . W + the shortcut's gradient (dense, compact stride-2 or (dy, mask)) as ONE GEMM with the previous block's bn3
    backward reduction in its epilogue (peclr_gemm_h)."""

    @staticmethod
    def forward(ctx, x, weight, conv, stats=None, link=None):
        ctx.save_for_backward(x)
        ctx.conv, ctx.link = conv, link
        planes = ctx.planes = conv.x6_group.planes(conv, x.dtype)
        n, cin, h, w = x.shape
        cmid = conv.out_channels
        x2 = x.permute(0, 2, 3, 1).reshape(n * h * w, cin)
        shift = _stat_shift_for(stats[0], cmid) if (stats and ROUTING.bn_stats_in_gemm) else None
        if shift is not None:
            y, partial, ns = _capi.gemm_h(x2, planes[0], cmid, tag="conv1x1_fwd", stat_shift=shift)
            stats[:] = [partial, ns, shift, stats[0]]
        else:
            y = _capi.gemm_h(x2, planes[0], cmid, tag="conv1x1_fwd")
        return y.view(n, h, w, cmid).permute(0, 3, 1, 2), x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gid):
        (x,) = ctx.saved_tensors
        conv, planes, link = ctx.conv, ctx.planes, ctx.link
        n, cin, h, w = x.shape
        cmid = conv.out_channels
        r = n * h * w
        gy = gy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dw = _wgrad_h(gy, x, conv, 1) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            a = gy.permute(0, 2, 3, 1).reshape(r, cmid)
            lazy = _take_lazy(gid)
            if lazy is None:
                kw = dict(addend=gid.to(x.dtype).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(r, cin))
            elif lazy[0] == "s2":
                kw = dict(addend=lazy[1], addend_s2=(h, w))
            else:
                kw = dict(addend=lazy[1].contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(r, cin), addend_mask=lazy[2])
            if link is not None and link[0].shape == x.shape and link[0].dtype == x.dtype and cin % 32 == 0:
                out, partial, ns = _capi.gemm_h(a, planes[1], cin, tag="conv1x1_dgrad_add", bn_bwd=link[:5], **kw)
                _note_bn_bwd(out, link, partial, ns)
            else:
                out = _capi.gemm_h(a, planes[1], cin, tag="conv1x1_dgrad_add", **kw)
            dx = out.view(n, h, w, cin).permute(0, 3, 1, 2)
        return dx, dw, None, None, None


def fork_conv1x1(conv: nn.Conv2d, x: Tensor, stats_for=None):
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
    if ok and _h_ok(conv, x):
        stats = [stats_for] if stats_for is not None else None
        out, identity = _ForkConvH.apply(x, conv.weight, conv, stats, _bn_link_of(x))
        identity._peclr_compact_ok = True         # its backward takes the shortcut's gradient compact / as (dy, mask)
        return _attach_stats(out, stats), identity
    if ok:
        stats = [stats_for] if stats_for is not None else None
        flags = []
        out, identity = _ForkConv1x1.apply(x, conv.weight, conv, stats, _bn_link_of(x), flags, _absmax_of(x))
        if _absmax_of(x) is not None:
            _tag_absmax(identity, _absmax_of(x))            # (the same values: the downsample convolution reads them too)
        if flags and flags[0]:
            identity._peclr_compact_ok = True     # a 1x1 / stride-2 shortcut may hand its input gradient over compact
        return _attach_stats(out, stats), identity
    return (conv(x, stats_for=stats_for) if isinstance(conv, Conv2d) else conv(x)), x     # (x has two consumers: no link)


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
        """(running_mean, running_var, num_batches_tracked, sync_shift) for the kernels.  The running statistics are
        NOT handed over while a checkpointed block is being re-run in the backward pass (the first run already moved
        them).  Synchronised statistics subtract a common shift -- the replicated running mean -- before summing: the
        first run keeps the copy it used, and the re-run gets that copy (the running mean has moved since) so that it
        reproduces the first run's statistics exactly and updates nothing."""
        sync = training and self.sync_group is not None
        if training and _RECOMPUTING:
            return None, None, None, (_ckpt_shift_of(self) if sync else None)
        shift = None
        if sync:
            shift = _ckpt_shift_of(self)         # (first run of a checkpointed block: the copy kept for its re-run)
            if shift is None:
                shift = self.running_mean.detach().clone()
        return self.running_mean, self.running_var, self.num_batches_tracked, shift

    def _takes_deferred_residual(self, x: Tensor) -> bool:
        """Can this layer's pass compute the shortcut's BatchNorm itself (peclr_bn2d_apply_res_bn)?  The plain fused pass only
        (not the stem's pooled form, not layer4's average-pooled tail)."""
        return bool(ROUTING.bn_shortcut_in_add and self.hip and self.affine and x.is_cuda and x.dim() == 4
                    and x.dtype in (torch.float32, torch.bfloat16, torch.float16)
                    and not (self.tail_avgpool and self.num_features % 32 == 0) and not self.default_pool)

    def forward(self, x: Tensor, residual: Optional[Tensor] = None, relu: Optional[bool] = None, consumer=None) -> Tensor:
        """consumer: the ONLY reader of the result.  For the BatchNorm of a shortcut (conv1x1 -> bn, no ReLU) it is the block's
        last FusedBatchNormAct2d, which adds the shortcut in its own pass (`_takes_deferred_residual`): the layer then only
        finishes its statistics and returns a placeholder carrying `_peclr_deferred = [x, scale_shift, relu]`: no apply pass, no
        output tensor.  A residual that is such a placeholder is computed on the fly (or materialised).  (A Conv2d consumer that
        applied the layer in its own operand path was built in round 5 and taken out in round 6: docs/history.md E.)"""
        relu = self.default_relu if relu is None else relu
        pool = self.default_pool and relu and residual is None
        if self.hip:
            global _DRAIN_QUEUED
            if _DRAIN_QUEUED and not _RECOMPUTING and not _NESTED_BACKWARD:
                # a forward pass outside any backward: the drain queued by the last backward pass never ran (that pass raised, and
                # autograd skips its final callbacks then) -- loops that do not call end_backward() would never queue one again
                _DRAIN_QUEUED = False
                end_backward()
            if not self.affine or (self.training and self.momentum is None):
                raise _capi.PeclrHipError("fused BatchNorm2d needs affine=True and a fixed momentum")
            # statistics its producer already summed (a GEMM epilogue, `conv_stats_for`), valid for THIS layer in training
            pre = getattr(x, "_peclr_bn_stats", None)
            if pre is not None and (pre[3] is not self or not (self.training or not self.track_running_stats)):
                pre = None
            pre = pre[:3] if pre is not None else None
            # a residual that is the placeholder of the shortcut's BatchNorm: computed inside this layer's pass, or written after all
            res_deferred = getattr(residual, "_peclr_deferred", None) if residual is not None else None
            if res_deferred is not None:
                res_deferred = tuple(res_deferred)
                if (res_deferred[2] or not self._takes_deferred_residual(x) or res_deferred[0].shape != x.shape
                        or res_deferred[0].dtype != x.dtype):
                    residual, res_deferred = _materialized(residual, res_deferred), None
            amax = [] if (ROUTING.x6_pair and x.dtype == torch.float32) else None
            if pool:
                y = _BN2dReluPool.apply(x, self.weight, self.bias, self, pre, amax)
                if amax:
                    _tag_absmax(y, amax[0])
                return y
            if self.tail_avgpool and relu and residual is not None and self.num_features % 32 == 0:
                return _BN2dAddReluAvgPool.apply(x, self.weight, self.bias, residual, self, pre)
            link = [] if (ROUTING.bn_bwd_in_gemm and torch.is_grad_enabled() and x.requires_grad) else None
            lazy_res = (ROUTING.lazy_residual_grad and relu and residual is not None and getattr(residual, "_peclr_compact_ok", False)
                        and torch.is_grad_enabled() and residual.requires_grad and self.num_features % 32 == 0)
            defer = None
            if consumer is not None and residual is None:
                if not relu and isinstance(consumer, FusedBatchNormAct2d) and consumer._takes_deferred_residual(x):
                    defer = []
            y = _BN2dAct.apply(x, self.weight, self.bias, residual, self, relu, pre, link, lazy_res, defer, res_deferred, amax)
            if amax:
                _tag_absmax(y, amax[0])         # max |y|, on the device: the "pair" GEMMs that read y derive its power of two from it
            if link:
                y._peclr_bn_link = link
            if defer:
                y._peclr_deferred = defer
            return y
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
        if (isinstance(m, Conv2d) and m.kernel_size == (7, 7) and m.stride == (2, 2) and m.padding == (3, 3) and m.in_channels == 3
                and m.out_channels == 64 and m.groups == 1 and m.bias is None and m.dilation == (1, 1)):
            m.hip_stem = enabled
            m.__dict__.pop("_stem_sets", None)
        if isinstance(m, Conv2d) and m.kernel_size in ((1, 1), (3, 3)) and m.stride in ((1, 1), (2, 2)):
            m.hip_gemm = enabled                # fp32 1x1 (where `_x6_pays`) and 3x3 stride-1 convolutions as in-tree GEMMs
            m.x6_group = None
    if enabled:                                 # their weights are split into bf16 planes once per step, all in one launch
        X6PackGroup([m for m in module.modules() if isinstance(m, Conv2d) and (m.hip_gemm or getattr(m, "hip_fork", False))])
    return n
