"""Optimiser plumbing of `BaseModel.configure_optimizers` (base_model.py:57-104).

The reference wires torch.optim.Adam -> pl_bolts LARSWrapper -> pl_bolts
LinearWarmupCosineAnnealingLR (pl_bolts==0.2.2; not vendored, not installed here: PARITY UNPINNED,
restated from its published behaviour, see oracle/peclr_oracle.py:lars_adam_step).

  LARSAdam                       torch.optim.Optimizer with LARSWrapper(Adam) semantics.
      fused=True  (HIP tensors): ONE fused multi-tensor HIP launch pair per optimiser step for all
                                 parameter groups (peclr_lars_sumsq_f32 + peclr_lars_adam_update_f32)
                                 -- no per-tensor norm launches, no host syncs.
      fused=False                the same update written with torch foreach ops (any device); used for
                                 CPU runs and as the comparison arm in tests.
  LARSWrapper(optimizer)         the reference's spelling: wraps an Adam built by the caller.
  DeviceLossScaler               precision=16: torch.amp.GradScaler's algorithm with the skip-on-inf decision taken
                                 on the device inside the fused step (`LARSAdam.attach_scaler`), so fp16 runs
                                 need no host sync per step and can be captured in hipGraphs.
  LinearWarmupCosineAnnealingLR  closed-form warm-up + cosine schedule, stepped per optimiser step.
"""
from __future__ import annotations

import math
from typing import List

import torch
from torch.optim import Optimizer
from torch.optim.lr_scheduler import LRScheduler

from . import _capi


def _is_dense(t: torch.Tensor) -> bool:
    """Dense in SOME dimension order (contiguous, channels_last, ...): storage has no gaps/overlaps."""
    if t.is_contiguous() or t.numel() <= 1:
        return True
    dims = sorted(range(t.dim()), key=lambda d: t.stride(d))
    expect = 1
    for d in dims:
        if t.size(d) == 1:
            continue
        if t.stride(d) != expect:
            return False
        expect *= t.size(d)
    return True


class _FusedWorkList:
    """Device-side work list over ALL parameter groups (rebuilt if the set of tensors changes)."""

    def __init__(self, params, grads, exp_avg, exp_avg_sq, group_of):
        dev = params[0].device
        n = len(params)
        self.key = tuple(t.data_ptr() for t in (*params, *grads, *exp_avg, *exp_avg_sq))
        ptrs = [t.data_ptr() for seq in (params, grads, exp_avg, exp_avg_sq) for t in seq]
        sizes = [p.numel() for p in params]
        chunk_tensor, chunk_offset, begin = [], [], [0]
        for t, sz in enumerate(sizes):
            for off in range(0, sz, _capi.OPT_CHUNK):
                chunk_tensor.append(t)
                chunk_offset.append(off)
            begin.append(len(chunk_tensor))
        self.n_tensors, self.n_chunks = n, len(chunk_tensor)
        self.ptrs = torch.tensor(ptrs, dtype=torch.int64, device=dev)
        self.sizes = torch.tensor(sizes, dtype=torch.int64, device=dev)
        self.chunk_tensor = torch.tensor(chunk_tensor, dtype=torch.int32, device=dev)
        self.chunk_offset = torch.tensor(chunk_offset, dtype=torch.int64, device=dev)
        self.begin = torch.tensor(begin, dtype=torch.int32, device=dev)
        self.group = torch.tensor(group_of, dtype=torch.int32, device=dev)
        self.norms_ws = torch.empty(2 * self.n_chunks, dtype=torch.float32, device=dev)


class DeviceLossScaler:
    """Dynamic loss scaling for precision=16 with torch.amp.GradScaler's algorithm and defaults (what
    Lightning 1.0.8's native-AMP plugin wraps around the reference's optimiser, peclr_training.py:78-79):

        backward on loss * scale;  at the optimiser step: g <- g / scale, and if any g is inf / nan the step
        is skipped and scale *= backoff_factor, else after `growth_interval` clean steps scale *= growth_factor.

    GradScaler.step() reads the inf flag back to the host.  Here the 16-byte state (`peclr_amp_state`: scale,
    found_inf, growth_tracker, good_steps) stays in device memory and the fused optimiser kernels act on it
    (peclr_lars_sumsq_amp_f32 / peclr_lars_adam_update_amp_f32 / peclr_amp_update): no sync, capturable.
    Use: `opt.attach_scaler(s)`; `s.scale(loss).backward()`; `opt.step()` -- there is no separate
    unscale_/step/update.  `state_dict()` has GradScaler's keys (Lightning's `native_amp_scaling_state`), so
    either class loads the other's checkpoint."""

    def __init__(self, device, init_scale: float = 2.0 ** 16, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000):
        if not (growth_factor > 1.0 and 0.0 < backoff_factor < 1.0 and growth_interval >= 1):
            raise ValueError("DeviceLossScaler: growth_factor > 1, 0 < backoff_factor < 1, growth_interval >= 1")
        if torch.device(device).type != "cuda":
            raise _capi.PeclrHipError("DeviceLossScaler lives in HIP device memory (CPU runs: torch.amp.GradScaler)")
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.state = torch.zeros(4, dtype=torch.int32, device=device)      # peclr_amp_state
        self._f = self.state.view(torch.float32)                            # words 0, 1: scale, found_inf
        self._f[0] = float(init_scale)

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        return loss * self._f[0]

    def get_scale(self) -> float:
        return float(self._f[0].item())

    def good_steps(self) -> int:
        return int(self.state[3].item())

    def set_good_steps(self, n: int):
        self.state[3] = int(n)

    def kernel_args(self):
        return (self.state, float(self.growth_factor), float(self.backoff_factor), int(self.growth_interval))

    def state_dict(self):
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(self.state[2].item())}

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self._f[0] = float(sd["scale"])
        self._f[1] = 0.0
        self.state[2] = int(sd["_growth_tracker"])


class LARSAdam(Optimizer):
    """Adam (torch defaults: betas (0.9, 0.999), eps 1e-8) with the LARSWrapper pre-step:

        for every param with a grad, if |p| != 0 and |g| != 0:
            trust = eta*|p| / (|g| + wd*|p| + lars_eps);  if clip: trust = min(trust / lr, 1)
            g <- (g + wd*p) * trust
        Adam step on g with weight_decay = 0.

    `lars=False` gives plain torch.optim.Adam (L2 weight decay) -- the non-"LARS" branch of
    configure_optimizers (base_model.py:99-100).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, lars=True, eta=0.02,
                 lars_eps=1e-8, clip=True, fused=None, write_back=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.lars, self.eta, self.lars_eps, self.clip = lars, eta, lars_eps, clip
        # True: leave the LARS-scaled gradient (g + wd*p)*trust in p.grad after the step, as the
        # reference's LARSWrapper does (it rewrites p.grad in place before Adam reads it); False
        # (default) keeps it in registers -- nothing on the training path reads .grad after the step
        self.write_back = bool(write_back) and lars
        first = self.param_groups[0]["params"][0]
        self.fused = first.is_cuda if fused is None else fused
        if self.fused and not first.is_cuda:
            raise _capi.PeclrHipError("LARSAdam(fused=True) needs HIP device parameters")
        self._fused_cache = {}
        # hipGraph support: per-step scalars live in device memory (`_hyper`), refreshed by
        # `prepare_step()` OUTSIDE the graph; the captured launch (`launch_only()`) only reads them
        self._hyper = self._hyper_host = None
        self._prepared = None
        self._amp = None

    def attach_scaler(self, scaler: "DeviceLossScaler"):
        """precision=16: from now on the gradients this optimiser sees are scale * g.  Every fused step
        unscales them in registers, is skipped ON THE DEVICE when one is inf / nan, and updates the scale.
        Adam's step count (bias corrections) becomes the device's count of steps actually taken; the host-side
        `state[p]["step"]` is refreshed from it in `state_dict()`."""
        if not self.fused:
            raise _capi.PeclrHipError("attach_scaler needs the fused HIP optimiser (CPU runs: torch.amp.GradScaler)")
        self._amp = scaler
        steps = [int(st["step"]) for st in self.state.values() if "step" in st]
        scaler.set_good_steps(max(steps) if steps else 0)
        return self

    def _amp_args(self):
        return None if self._amp is None else self._amp.kernel_args()

    def state_dict(self):
        if self._amp is not None:                      # skipped steps were counted on the host, not on the device
            n = self._amp.good_steps()
            for st in self.state.values():
                if "step" in st:
                    st["step"] = n
        return super().state_dict()

    def _lars_mode(self) -> int:
        return (2 if self.write_back else 1) if self.lars else 0

    def load_state_dict(self, state_dict):
        """Restored moments are new tensors: drop the cached device work list (its pointer table) so the
        next step rebuilds it.  A captured hipGraph keeps its own work list alive; re-point that one with
        `repoint_worklist()` after loading (Trainer.resume runs before any capture)."""
        super().load_state_dict(state_dict)
        self._fused_cache.clear()
        self._prepared = None
        if self._amp is not None:
            steps = [int(st["step"]) for st in self.state.values() if "step" in st]
            self._amp.set_good_steps(max(steps) if steps else 0)

    def _prepare(self, group):
        """Lazy state init + step count for one group; returns (params, grads, m, v, step)."""
        params = [p for p in group["params"] if p.grad is not None]
        for p in params:
            st = self.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
        grads = [p.grad for p in params]
        m = [self.state[p]["exp_avg"] for p in params]
        v = [self.state[p]["exp_avg_sq"] for p in params]
        return params, grads, m, v, (self.state[params[0]]["step"] if params else 0)

    # ---- hipGraph-capturable split of step(): host part / device part
    @torch.no_grad()
    def prepare_step(self):
        """Host side of one fused step: advance the step counters and stage lr / weight decay / bias
        corrections into device memory.  Call before every graph replay (and before capturing)."""
        if not self.fused:
            raise _capi.PeclrHipError("prepare_step()/launch_only() need the fused HIP optimiser")
        prepared = [(g, *self._prepare(g)) for g in self.param_groups]
        prepared = [t for t in prepared if t[1]]
        if len(prepared) > 8 or len({(tuple(g["betas"]), g["eps"], st) for g, _, _, _, _, st in prepared}) != 1:
            raise _capi.PeclrHipError("graph mode needs <= 8 parameter groups with common betas / eps / step")
        g0, step = prepared[0][0], prepared[0][5]
        b1, b2 = g0["betas"]
        vals = [0.0] * 18
        for i, t in enumerate(prepared):
            vals[i], vals[8 + i] = float(t[0]["lr"]), float(t[0]["weight_decay"])
        vals[16], vals[17] = 1.0 - b1 ** step, 1.0 - b2 ** step
        if self._hyper is None:
            self._hyper = torch.zeros(18, dtype=torch.float32, device=prepared[0][1][0].device)
        # pageable source: the values are staged before copy_ returns, so the host may run ahead of
        # the GPU by any number of steps without racing on a shared pinned buffer
        self._hyper.copy_(torch.tensor(vals, dtype=torch.float32))
        self._prepared = prepared

    @torch.no_grad()
    def launch_only(self, reuse_worklist: bool = False):
        """Device side of the step prepared by `prepare_step()`: the two kernel launches, with every
        per-step scalar read from device memory -- safe to capture in a hipGraph and replay.
        reuse_worklist: launch with the existing device-side pointer table as is (inside a capture the
        gradients live at new addresses that are only known afterwards -> `repoint_worklist()`)."""
        self._step_fused(self._prepared, device_hyper=self._hyper, reuse=reuse_worklist)

    @torch.no_grad()
    def repoint_worklist(self):
        """Rewrite the device-side pointer table from the CURRENT param / grad / moment addresses (same
        tensors, same order, same sizes as when the work list was built)."""
        wl = self._fused_cache["all"]
        groups = [[p for p in g["params"] if p.grad is not None] for g in self.param_groups]
        params = [p for ps in groups for p in ps]
        if len(params) != wl.n_tensors:
            raise _capi.PeclrHipError("repoint_worklist: the set of parameters with gradients changed")
        seqs = (params, [p.grad for p in params], [self.state[p]["exp_avg"] for p in params],
                [self.state[p]["exp_avg_sq"] for p in params])
        for p_, g_ in zip(params, seqs[1]):
            if g_.stride() != p_.stride() or g_.dtype != torch.float32:
                raise _capi.PeclrHipError("repoint_worklist: grad layout differs from the parameter's")
        wl.ptrs.copy_(torch.tensor([t.data_ptr() for seq in seqs for t in seq], dtype=torch.int64))
        wl.key = tuple(t.data_ptr() for seq in seqs for t in seq)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        prepared = [(g, *self._prepare(g)) for g in self.param_groups]
        prepared = [t for t in prepared if t[1]]
        if not prepared:
            return loss
        uniform = len({(tuple(g["betas"]), g["eps"], st) for g, _, _, _, _, st in prepared}) == 1
        if self.fused and uniform and len(prepared) <= 8:
            self._step_fused(prepared)  # ONE launch pair for every parameter group
            return loss
        if self._amp is not None:
            raise _capi.PeclrHipError("a DeviceLossScaler needs ONE fused launch for all parameter groups (<= 8 "
                                      "groups with common betas / eps): the scale is updated once per step")
        for g, params, grads, m, v, step in prepared:
            b1, b2 = g["betas"]
            bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
            if self.fused:
                self._step_fused([(g, params, grads, m, v, step)])
            else:
                self._step_foreach(params, grads, m, v, float(g["lr"]), b1, b2, g["eps"], g["weight_decay"], bc1,
                                   bc2)
        return loss

    # ---- HIP: two launches per optimiser step (sum of squares; update), all groups at once
    def _step_fused(self, prepared, device_hyper=None, reuse=False):
        if reuse:
            wl = self._fused_cache["all"]
            g0, step = prepared[0][0], prepared[0][5]
            b1, b2 = g0["betas"]
            _capi.lars_adam_step(wl.ptrs, wl.sizes, wl.n_tensors, wl.chunk_tensor, wl.chunk_offset, wl.begin,
                                 wl.group, wl.n_chunks, wl.norms_ws, [0.0] * len(prepared), [0.0] * len(prepared), b1,
                                 b2, g0["eps"], 1.0, 1.0, self._lars_mode(), self.eta, self.lars_eps, self.clip,
                                 device_hyper=device_hyper, amp=self._amp_args())
            return
        params = [p for t in prepared for p in t[1]]
        grads = [x for t in prepared for x in t[2]]
        m = [x for t in prepared for x in t[3]]
        v = [x for t in prepared for x in t[4]]
        # The kernel walks raw storage, so every tensor must be dense (any permutation: NCHW or
        # channels_last) and grad / moments must share the parameter's strides.
        for p_, g_, m_, v_ in zip(params, grads, m, v):
            if p_.dtype != torch.float32 or g_.dtype != torch.float32:
                raise _capi.PeclrHipError("LARSAdam(fused): parameters and grads must be fp32")
            if not _is_dense(p_):
                raise _capi.PeclrHipError("LARSAdam(fused): parameters must be dense (non-overlapping, no gaps)")
            if g_.stride() != p_.stride() or m_.stride() != p_.stride() or v_.stride() != p_.stride():
                raise _capi.PeclrHipError("LARSAdam(fused): grad / exp_avg / exp_avg_sq must share the parameter's "
                                          f"strides (param {tuple(p_.stride())}, grad {tuple(g_.stride())})")
        # moments included: load_state_dict (or anything else that swaps exp_avg / exp_avg_sq) must not leave
        # the device-side pointer table aimed at the old, possibly freed, buffers
        key = tuple(t.data_ptr() for t in (*params, *grads, *m, *v))
        wl = self._fused_cache.get("all")
        if wl is None or wl.key != key:
            group_of = [gi for gi, t in enumerate(prepared) for _ in t[1]]
            wl = self._fused_cache["all"] = _FusedWorkList(params, grads, m, v, group_of)
        g0, step = prepared[0][0], prepared[0][5]
        b1, b2 = g0["betas"]
        _capi.lars_adam_step(wl.ptrs, wl.sizes, wl.n_tensors, wl.chunk_tensor, wl.chunk_offset, wl.begin, wl.group,
                             wl.n_chunks, wl.norms_ws, [float(t[0]["lr"]) for t in prepared],
                             [float(t[0]["weight_decay"]) for t in prepared], b1, b2, g0["eps"], 1.0 - b1 ** step,
                             1.0 - b2 ** step, self._lars_mode(), self.eta, self.lars_eps, self.clip,
                             device_hyper=device_hyper, amp=self._amp_args())

    # ---- torch foreach restatement (any device)
    def _step_foreach(self, params, grads, m, v, lr, b1, b2, eps, wd, bc1, bc2):
        if self.lars:
            p_norm = torch.stack(torch._foreach_norm(params))
            g_norm = torch.stack(torch._foreach_norm(grads))
            trust = self.eta * p_norm / (g_norm + p_norm * wd + self.lars_eps)
            if self.clip:
                trust = torch.clamp(trust / lr, max=1.0) if lr > 0 else torch.ones_like(trust)
            active = (p_norm != 0) & (g_norm != 0)
            trust = torch.where(active, trust, torch.ones_like(trust))
            wds = torch.where(active, torch.full_like(trust, wd), torch.zeros_like(trust))
            g_eff = torch._foreach_mul(params, list(wds.unbind()))
            torch._foreach_add_(g_eff, grads)
            torch._foreach_mul_(g_eff, list(trust.unbind()))
            if self.write_back:
                torch._foreach_copy_(grads, g_eff)
        elif wd != 0:
            g_eff = torch._foreach_add(grads, params, alpha=wd)
        else:
            g_eff = list(grads)
        torch._foreach_mul_(m, b1)
        torch._foreach_add_(m, g_eff, alpha=1 - b1)
        torch._foreach_mul_(v, b2)
        torch._foreach_addcmul_(v, g_eff, g_eff, value=1 - b2)
        denom = torch._foreach_sqrt(v)
        torch._foreach_div_(denom, math.sqrt(bc2))
        torch._foreach_add_(denom, eps)
        torch._foreach_addcdiv_(params, m, denom, value=-lr / bc1)


def LARSWrapper(optimizer: Optimizer, eta: float = 0.02, clip: bool = True, eps: float = 1e-8) -> LARSAdam:
    """The reference's call shape `LARSWrapper(torch.optim.Adam(groups, lr=...))` (base_model.py:91):
    returns a LARSAdam over the SAME parameter groups and hyper-parameters."""
    if not isinstance(optimizer, torch.optim.Adam):
        raise TypeError("LARSWrapper here wraps torch.optim.Adam only (what the reference passes)")
    groups = [{k: v for k, v in g.items() if k in ("params", "lr", "betas", "eps", "weight_decay")}
              for g in optimizer.param_groups]
    return LARSAdam(groups, lars=True, eta=eta, lars_eps=eps, clip=clip, write_back=True)  # p.grad rewritten, as there


class LinearWarmupCosineAnnealingLR(LRScheduler):
    """Linear warm-up from `warmup_start_lr` to the base lr over `warmup_epochs` steps, then cosine
    annealing to `eta_min` at `max_epochs` (closed form of the pl_bolts scheduler; "epochs" are
    optimiser steps because the reference registers it with interval="step", base_model.py:102)."""

    def __init__(self, optimizer, warmup_epochs: int, max_epochs: int, warmup_start_lr: float = 0.0,
                 eta_min: float = 0.0, last_epoch: int = -1):
        self.warmup_epochs, self.max_epochs = warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self) -> List[float]:
        t = self.last_epoch
        out = []
        for base in self.base_lrs:
            if t < self.warmup_epochs:
                lr = self.warmup_start_lr + t * (base - self.warmup_start_lr) / max(self.warmup_epochs - 1, 1)
            else:
                lr = self.eta_min + 0.5 * (base - self.eta_min) * (
                    1 + math.cos(math.pi * (t - self.warmup_epochs) / max(self.max_epochs - self.warmup_epochs, 1)))
            out.append(lr)
        return out
