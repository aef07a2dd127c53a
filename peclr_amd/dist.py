"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm)
across xGMI; gloo on CPU for the tests.

The reference is single-device (`Trainer(gpus="0")`, peclr_training.py:73-81; no collective call
anywhere), so the multi-GPU semantics are defined here:

  p ranks x N_local pairs  ==  the single-device reference on the concatenated batch of p*N_local
  pairs (same loss, same summed parameter gradients), up to BatchNorm statistics, which stay
  per-rank (local BN; the per-rank step is then exactly the reference's step on that rank's rows
  as far as BN is concerned).

Exchange steps per optimisation step:
  1. all-gather of the projected embeddings z [2*N_local,128] fp32 (128 KiB/rank at N_local=128)
     and of one packed vector [row_lse | partial loss] (1 KiB/rank): latency-bound, two small
     RCCL all-gathers (`ops.ntxent`).
  2. SUM all-reduce of the parameter gradients (RN-50: ~94 MB fp32), in a few LARGE flat buckets:
     xGMI is point-to-point (7 links x ~153 GB/s per GPU), ring collectives are bound by one link
     and RCCL needs big messages to stripe across links, so the bucket size defaults to 32 MiB
     (ResNet-50: head + layer4 in two buckets, layer3 in one, and a last 6 MB bucket for layer2 /
     layer1 / stem -- the only one whose all-reduce cannot hide behind remaining backward work) and
     buckets are launched from autograd hooks as soon as their last gradient lands (overlap with the
     rest of the backward).  `encoder.final_layer.*` never receives a gradient and is skipped.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size(group=None) -> int:
    return dist.get_world_size(group) if is_initialized() else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if is_initialized() else 0


def init_from_env(backend: Optional[str] = None) -> int:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).
    Returns the local rank.  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        # PECLR_SHARE_DEVICE=1 (tests only): every rank on GPU 0, to rehearse the N > 1 path on a
        # single-GPU box -- RCCL refuses two ranks per device, so pair it with PECLR_DIST_BACKEND=gloo
        local = 0 if os.environ.get("PECLR_SHARE_DEVICE") == "1" else local
        torch.cuda.set_device(local)
    if world > 1 and not is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("PECLR_DIST_BACKEND") or \
            ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world, **kwargs)
    return local


# Test hook: issue every collective even on a one-rank group (where each is the identity).  A single-GPU box cannot host
# two RCCL ranks (RCCL refuses two ranks per device), so this is how the RCCL branches -- all_gather_into_tensor, the
# asynchronous bucket all-reduce between hipGraph replays -- get executed with a live RCCL communicator there
# (tests/test_dist_rccl_gpu.py).  Off in production.
FORCE_COLLECTIVES = False


def collectives_active(group=None) -> bool:
    """More than one rank -- or the one-rank test hook above with a live process group."""
    return world_size(group) > 1 or (FORCE_COLLECTIVES and dist.is_initialized())


def all_gather_cat(t: Tensor, group=None) -> Tensor:
    """[rows, ...] on every rank -> [world*rows, ...], rank-major.  Not differentiable (the callers
    carry their own closed-form backward)."""
    world = world_size(group)
    if world == 1 and not (FORCE_COLLECTIVES and dist.is_initialized()):
        return t
    t = t.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    if dist.get_backend(group) == "gloo" and t.is_cuda:  # gloo has no fused all-gather for device tensors
        dist.all_gather(list(out.chunk(world)), t, group=group)
    else:
        dist.all_gather_into_tensor(out, t, group=group)
    return out


def assert_uniform(value: int, group=None, device=None, what: str = "value"):
    """Raise on EVERY rank if `value` is not the same on all ranks of `group` (one 2-element MAX
    all-reduce of [v, -v])."""
    if world_size(group) == 1:
        return
    if device is None or dist.get_backend(group) == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([value, -value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0]), -int(t[1])
    if hi != lo:
        raise RuntimeError(f"{what} differs across data-parallel ranks (min {lo}, max {hi}, this rank {value}): "
                           "every rank must hold the same number; drop or pad the ragged batch")


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        pad = lambda n: (n + 63) // 64 * 64  # 256-byte slots: every view stays 16-byte aligned, so the
        numel = sum(pad(p.numel()) for p in params)  # fused optimiser keeps its 16-byte vector path
        self.flat = torch.zeros(numel, device=params[0].device, dtype=params[0].dtype)
        self.views, o = [], 0
        for p in params:
            # same sizes AND strides as the parameter (dense NCHW or channels_last): autograd's
            # gradient-layout contract, and what the fused optimiser walks raw storage with
            self.views.append(self.flat[o:o + p.numel()].as_strided(p.size(), p.stride()))
            o += pad(p.numel())
        self.pending = len(params)
        self.handle = None


class GradReducer:
    """Bucketed SUM all-reduce of gradients, overlapped with backward.

    Every parameter's `.grad` is a view into its bucket's flat buffer, so autograd accumulates
    straight into the communication buffer (no pack/unpack copies).  `p.register_post_accumulate_
    grad_hook` counts arrivals; the bucket's all-reduce is issued asynchronously when the last one
    lands.  Call `prepare()` before each backward that ends an accumulation window, `finish()` before
    the optimiser step, and `zero_grad()` instead of `optimizer.zero_grad()`.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 32 << 20, stage_of=None):
        """stage_of (optional): parameter -> int.  A bucket never mixes stages, so that a backward pass cut
        into stages (Trainer.capture_split_graphs: head + layer4 | the rest) can all-reduce a finished
        stage's buckets while the next stage is still computing."""
        self.group = group
        self.world = world_size(group)
        params = [p for p in params if p.requires_grad]
        stage_of = stage_of or (lambda p: 0)
        # buckets in REVERSE registration order: the last layers' grads are ready first
        groups: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes = [], 0
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype or stage_of(p) != stage_of(cur[0])):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        # a small tail (the last few parameters of a stage) rides with its predecessor instead of paying for an
        # all-reduce of its own: every collective costs a fixed latency over xGMI
        size = lambda g: sum(p.numel() * p.element_size() for p in g)  # noqa: E731
        merged: List[List[torch.nn.Parameter]] = []
        for g in groups:
            if (merged and size(g) < bucket_bytes // 8 and g[0].dtype == merged[-1][0].dtype
                    and stage_of(g[0]) == stage_of(merged[-1][0])):
                merged[-1] = merged[-1] + g
            else:
                merged.append(g)
        self.buckets: List[_Bucket] = [_Bucket(g) for g in merged]
        for b in self.buckets:
            b.stage = stage_of(b.params[0])
        self._armed = False
        self._owner = {}
        for b in self.buckets:
            for p, v in zip(b.params, b.views):
                p.grad = v
                self._owner[id(p)] = b
                p.register_post_accumulate_grad_hook(self._hook)

    def _hook(self, p):
        if not self._armed or not self._reduces():
            return
        b = self._owner[id(p)]
        b.pending -= 1
        if b.pending == 0:
            b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _reduces(self) -> bool:
        return self.world > 1 or (FORCE_COLLECTIVES and dist.is_initialized())

    def prepare(self, unused: Iterable[torch.nn.Parameter] = ()):
        """Arm the hooks for the coming backward.  `unused`: parameters that will NOT receive a
        gradient in it (e.g. encoder.final_layer.*), so their buckets do not wait for them."""
        self._armed = True
        skip = {id(p) for p in unused}
        for b in self.buckets:
            b.pending = sum(1 for p in b.params if id(p) not in skip)
            b.handle = None

    def finish(self):
        """Wait for every in-flight bucket; reduce any bucket whose hooks never completed."""
        if self._reduces():
            for b in self.buckets:
                if b.handle is None:
                    b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for b in self.buckets:
                b.handle.wait()
                b.handle = None
        self._armed = False

    def all_reduce_now(self):
        """Reduce every bucket (no hooks involved): the path used after a captured backward, whose
        gradients were copied into the buckets."""
        if self._reduces():
            handles = [dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets]
            for h in handles:
                h.wait()

    def launch(self, buckets: Iterable[_Bucket]) -> list:
        """Start the SUM all-reduce of some buckets asynchronously (RCCL's own stream picks up after the work
        already queued on the current stream); returns handles to `wait()` on before the optimiser step."""
        if not self._reduces():
            return []
        return [dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in buckets]

    def zero_grad(self):
        for b in self.buckets:
            b.flat.zero_()
            for p, v in zip(b.params, b.views):
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    p.grad = v  # re-attach if something replaced the view


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
