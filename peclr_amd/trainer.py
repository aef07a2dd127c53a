"""Minimal training loop with the Lightning 1.0.8 semantics the reference's step relies on
(peclr_training.py:73-81,96): loss / accumulate_grad_batches, backward per
micro-batch, optimizer.step + zero_grad + scheduler.step every K-th batch (interval="step"),
epoch-end hooks, top-k checkpoints on `checkpoint_saving_loss`.  One process per GPU; gradients
are SUM-all-reduced over RCCL/xGMI by `dist.GradReducer`, overlapped with backward.

Not a Lightning re-implementation: no loggers, no callbacks API, no dataloader management.
"""
from __future__ import annotations

import contextlib
import gc
import os
from typing import Callable, Dict, Iterable, List, Optional

import torch

from . import dist as pdist
from .port import save_checkpoint


def _flag_last(iterable):
    """(item, is_last) pairs with one item of look-ahead (Lightning's `is_final_batch`)."""
    it = iter(iterable)
    try:
        prev = next(it)
    except StopIteration:
        return
    for cur in it:
        yield prev, False
        prev = cur
    yield prev, True


class Trainer:
    def __init__(self, max_epochs: int = 1, accumulate_grad_batches: int = 1, precision: str = "fp32",
                 checkpoint_dir: Optional[str] = None, save_top_k: int = 1, process_group=None,
                 bucket_bytes: int = 32 << 20, channels_last: bool = False, grad_buckets=None,
                 sync_batchnorm: bool = False, hip_graph: bool = False, activation_checkpointing: bool = False,
                 overlap_wgrad: bool = False):
        self.max_epochs = max_epochs
        # True: the backbone's weight gradients are computed on a second HIP stream (peclr_amd.bn2d), next to the
        # BatchNorm / residual glue of the layers below instead of in front of it; joined after every backward.
        self.overlap_wgrad = overlap_wgrad
        # True: residual blocks of the encoder keep only their input for backward and recompute the rest there
        # (peclr_amd.resnet.set_activation_checkpointing): for batches / resolutions whose activations do not fit
        self.activation_checkpointing = activation_checkpointing
        self.accumulate_grad_batches = accumulate_grad_batches
        # "fp32" | "bf16" | 16 / "16" / "fp16".  16 is the reference's default (training_config.json:9,
        # peclr_training.py:78-79: Lightning native AMP = fp16 autocast + a dynamic GradScaler); it is
        # reproduced with torch.amp.GradScaler: loss scaled before backward, gradients unscaled (after the
        # all-reduce) and checked before the step, step skipped and scale halved on inf/nan.
        self.precision = {16: "fp16", "16": "fp16", 32: "fp32", "32": "fp32"}.get(precision, precision)
        if self.precision not in ("fp32", "bf16", "fp16"):
            raise ValueError(f"precision {precision!r}: expected 'fp32', 'bf16' or 16/'fp16'")
        self._scaler = None
        self._uniform_n = None
        self.checkpoint_dir = checkpoint_dir
        self.save_top_k = save_top_k
        self.process_group = process_group
        self.bucket_bytes = bucket_bytes
        self.channels_last = channels_last
        self.grad_buckets = grad_buckets  # None: only when world_size > 1; True: always (flat grad buffers)
        # True: BatchNorm statistics (backbone and head) are those of the GLOBAL batch, so N ranks x B
        # samples compute exactly what one device with N*B samples computes (costs two small
        # all-reduces per BN layer per step).  False: per-rank statistics, like DDP without SyncBatchNorm.
        self.sync_batchnorm = sync_batchnorm
        # True: fit() captures the training step on the first batch (one hipGraph; a forward graph + one
        # backward graph per stage around the collectives when gradients live in all-reduce buckets; one
        # micro-batch graph with accumulate_grad_batches > 1) and replays it for every later batch of
        # the same shapes; other shapes (a ragged last batch) run eagerly.
        self.hip_graph = hip_graph
        self._graph_sig = None
        self.world_size = pdist.world_size(process_group)
        self.global_step = 0
        self.current_epoch = 0
        self._saved: List[tuple] = []
        self.model = self.optimizer = self.scheduler = self.reducer = None
        self._graphs_alive: List[torch.cuda.CUDAGraph] = []    # every graph this trainer captured, until `close()`


    # ---- hipGraph lifetime.  Destroying a hipGraph (or freeing graph-pool memory) while a stream capture is in progress
    # aborts the process on this ROCm build (seen as `Fatal Python error: Aborted` when a cyclic garbage collection ran
    # inside a capture and found a dead CUDAGraph).  So: (i) garbage is collected right BEFORE a capture starts, while
    # nothing is being recorded; (ii) the collector is off for the duration of the capture (objects that become garbage in
    # there wait for the next collection after it); (iii) the trainer keeps a strong reference to every graph it captured
    # until `close()`, so that re-capturing (another batch shape, a new accumulation window) never drops the last
    # reference to an executable graph from inside Python code that may itself be running under a capture.
    @contextlib.contextmanager
    def _capturing(self, graph: "torch.cuda.CUDAGraph", **kw):
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(graph, **kw):
                yield graph
        finally:
            if was_enabled:
                gc.enable()
        self._graphs_alive.append(graph)

    def _release_superseded(self):
        """A new capture (another batch shape, another model, a second `attach`) supersedes what this trainer captured before:
        release those graphs and the pool memory they pin NOW -- outside any capture, device idle -- instead of keeping every
        generation alive until `close()`."""
        if self._graphs_alive:
            sig = self._graph_sig
            self.close()
            self._graph_sig = sig              # (`_graph_step` sets it before it captures)

    def close(self):
        """Release every captured graph (and the graph-pool memory they pin): outside any capture, device idle."""
        if torch.cuda.is_available():
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("Trainer.close() inside a stream capture")
            torch.cuda.synchronize()
        for name in ("_graph", "_graph_a", "_graph_b", "_graph_b2", "_graph_bs", "_static_out", "_static_grads", "_split_seams",
                     "_split_stages", "_split_z", "_split_dz", "_micro_src", "_graph_worklist"):
            if hasattr(self, name):
                setattr(self, name, None)
        self._graphs_alive.clear()
        self._graph_sig = None
        gc.collect()

    # ---- setup
    def attach(self, model):
        if getattr(self, "_graphs_alive", None):       # re-attach: the graphs of the previous model go (see `_release_superseded`)
            self.close()
        model.trainer = self
        model.process_group = self.process_group
        model.setup("fit")
        pdist.broadcast_module_state(model, 0, self.process_group)
        self.model = model
        if self.activation_checkpointing:
            from .resnet import set_activation_checkpointing

            if set_activation_checkpointing(model.encoder, True) == 0:
                raise RuntimeError("activation_checkpointing: the encoder has no peclr_amd.resnet blocks")
        if self.overlap_wgrad:
            from . import bn2d as _bn2d

            _bn2d.enable_wgrad_overlap(True)
        if self.sync_batchnorm and self.world_size > 1:
            self._enable_sync_batchnorm(model)
        if self.world_size > 1 or self.grad_buckets:
            self.reducer = pdist.GradReducer(model.parameters(), self.process_group, self.bucket_bytes,
                                             stage_of=self._backward_stage_of(model))
        (self.optimizer,), (sched,) = model.configure_optimizers()
        self.scheduler = sched["scheduler"]
        self._unused = [p for n, p in model.named_parameters() if "final_layer" in n]
        return self

    @staticmethod
    def _backward_stage_of(model):
        """Backward stage of every parameter for the staged split backward: 0 = projection head + layer4 (+ the
        unused final_layer), 1 = layer3, 2 = layer2 + layer1 + stem (cuts at ResNetModel.SEAMS).  None when the
        encoder is not the in-tree wrapper."""
        from .encoder import ResNetModel

        enc = getattr(model, "encoder", None)
        if not isinstance(enc, ResNetModel):
            return None
        stage = {}
        bounds = sorted(ResNetModel.SEAMS, reverse=True)            # (7, 6)
        for k, hi in enumerate(bounds):                             # features[bounds[k+1] : bounds[k]] is stage k + 1
            lo = bounds[k + 1] if k + 1 < len(bounds) else 0
            for p in enc.features[lo:hi].parameters():
                stage[id(p)] = k + 1
        return lambda p: stage.get(id(p), 0)

    def _enable_sync_batchnorm(self, model):
        import torch.distributed as td

        from .bn2d import FusedBatchNormAct2d

        group = self.process_group if self.process_group is not None else td.group.WORLD
        for name, m in model.named_modules():
            if isinstance(m, FusedBatchNormAct2d):
                if not m.hip:
                    raise RuntimeError(f"sync_batchnorm needs the fused HIP BatchNorm ({name}): call "
                                       "enable_hip_batchnorm(model.encoder) on an NHWC encoder first")
                m.sync_group = group
            elif isinstance(m, torch.nn.BatchNorm2d):
                raise RuntimeError(f"sync_batchnorm: {name} is a stock BatchNorm2d; build the encoder with "
                                   "peclr_amd.resnet (FusedBatchNormAct2d)")
        model.sync_bn_group = group  # projection-head BatchNorm1d (ops.head_align)

    def _autocast(self):
        """Context of every forward pass the trainer runs (also arms the side-stream weight gradients for it: the
        trainer follows each of its backward passes with `_join_wgrad`)."""
        if self.overlap_wgrad:
            from . import bn2d as _bn2d

            _bn2d.arm_wgrad_overlap()
        if self.precision == "fp32":
            return contextlib.nullcontext()
        dev = "cuda" if next(self.model.parameters()).is_cuda else "cpu"
        return torch.autocast(dev, dtype=torch.bfloat16 if self.precision == "bf16" else torch.float16)

    def _grad_scaler(self):
        """fp16 only: the dynamic loss scaler of native AMP (created on first use, saved in checkpoints).
        With the fused HIP optimiser it is a `DeviceLossScaler` attached to it -- same algorithm and checkpoint
        keys as torch's GradScaler, decision taken on the device (no host sync; hipGraph-capturable); on other
        devices torch's own GradScaler around the foreach optimiser."""
        if self.precision != "fp16":
            return None
        if self._scaler is None:
            p = next(self.model.parameters())
            if getattr(self.optimizer, "fused", False) and hasattr(self.optimizer, "attach_scaler"):
                from .optim import DeviceLossScaler

                self._scaler = DeviceLossScaler(p.device)
                self.optimizer.attach_scaler(self._scaler)
            else:
                self._scaler = torch.amp.GradScaler("cuda" if p.is_cuda else "cpu")
        return self._scaler

    def _device_scaler(self):
        from .optim import DeviceLossScaler

        return isinstance(self._grad_scaler(), DeviceLossScaler)

    def _scaled(self, loss):
        """precision=16: loss * scale (a device scalar: nothing is read back), else the loss itself."""
        scaler = self._grad_scaler()
        return loss if scaler is None else scaler.scale(loss)

    def _check_uniform_batch(self, batch):
        """N > 1: the all-gather of the embeddings has a fixed shape and the positive-pair index map uses
        the local pair count on global rows, so every rank must hold the same number of pairs.  One tiny MAX
        all-reduce of [n, -n] on EVERY eager micro-batch and on every rank (a rank-local "only when my count
        changed" shortcut would leave the rank with the short batch alone in the collective): a ragged batch
        that differs across ranks raises on every rank instead of hanging or silently mis-pairing positives."""
        if self.world_size == 1:
            return
        n = int(batch["transformed_image1"].shape[0])
        pdist.assert_uniform(n, self.process_group, batch["transformed_image1"].device, "pairs per rank")
        self._uniform_n = n

    def _ranks_agree_on_replay(self, fits: bool, is_final_batch: bool) -> bool:
        """Graph mode at N > 1: replay only if the batch fits the captured shapes on EVERY rank.  Decided with one
        MIN all-reduce where a ragged batch can occur (the epoch's final batch: every rank is at its final batch, so
        the collective is symmetric); elsewhere a batch that does not fit is an error raised before any collective.
        That error is raised by the rank that sees the odd batch ONLY: the other ranks replay, enter their next collective
        and are ended by RCCL's watchdog timeout (torch.distributed's default: 10 minutes) -- a loud failure, not a silent one,
        but a slow one; a loader with `drop_last=True` / equal shards never gets here, and deciding every batch with a
        collective would put a host synchronisation into every replayed step."""
        if self.world_size == 1 or self.reducer is None:
            return fits
        if not is_final_batch:
            if not fits:
                raise RuntimeError("a batch whose shapes differ from the captured graph's arrived before the epoch's final "
                                   "batch: with data parallelism only the final batch of an epoch may be ragged")
            return True
        dev = next(self.model.parameters()).device
        flag = torch.tensor([int(fits)], device="cpu" if torch.distributed.get_backend(self.process_group) == "gloo" else dev,
                            dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=self.process_group)
        return bool(int(flag))

    def zero_grad(self):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    # ---- one micro-batch; returns the step's output dict
    def training_micro_step(self, batch: Dict[str, torch.Tensor], batch_idx: int,
                            is_final_batch: bool = False) -> Dict[str, torch.Tensor]:
        """Lightning 1.0.8 accumulation: the loss is divided by k for every micro-batch; the optimiser
        steps when the window is full OR on the epoch's final batch (`should_accumulate = not
        (accumulation_done or is_final_batch)`), so a trailing partial window is applied, not leaked into
        the next epoch."""
        k = self.accumulate_grad_batches
        last = (batch_idx + 1) % k == 0 or is_final_batch
        self._check_uniform_batch(batch)
        if self.reducer is not None and last:
            self.reducer.prepare(self._unused)
        scaler = self._grad_scaler()
        with self._autocast():
            out = self.model.training_step(batch, batch_idx)
        loss = out["loss"] / k
        (scaler.scale(loss) if scaler is not None else loss).backward()
        self._join_wgrad()
        if last:
            if self.reducer is not None:
                self.reducer.finish()
            if scaler is not None and not self._device_scaler():
                scaler.step(self.optimizer)   # unscale, inf/nan check, step unless one was found
                scaler.update()
            else:
                self.optimizer.step()         # fp16 + fused: unscale / check / skip / scale update inside
            self.zero_grad()
            self.scheduler.step()
            self.global_step += 1
        return {key: v.detach() for key, v in out.items()}

    def _join_wgrad(self):
        """After a backward pass: the current stream waits for the side stream's weight gradients, and the gradient
        hand-over side channels of the fused backbone are emptied (bn2d.end_backward)."""
        from . import bn2d as _bn2d

        if self.overlap_wgrad:
            _bn2d.wgrad_join(self.reducer._hook if self.reducer is not None else None)
        _bn2d.end_backward()

    def _no_fp16_graphs(self):
        if self.precision == "fp16" and not self._device_scaler():
            raise RuntimeError("hipGraph capture with precision=16 needs the fused HIP optimiser (LARSAdam(fused=True)): "
                               "torch's GradScaler.step decides on the host whether to step")

    @staticmethod
    def _clone_batch(batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Static copy of a batch for graph replay.  When the two views arrive as halves of one stacked
        tensor (`transformed_images`), the copy keeps that relation (no per-step concatenation)."""
        out = {k: v.clone() for k, v in batch.items() if not (k.startswith("transformed_image") and "transformed_images" in batch)}
        if "transformed_images" in batch:
            stacked = batch["transformed_images"].clone(memory_format=torch.preserve_format)
            n = stacked.shape[0] // 2
            out.update(transformed_images=stacked, transformed_image1=stacked[:n], transformed_image2=stacked[n:])
        return out

    @staticmethod
    def _load_static(static: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor]):
        stacked = "transformed_images" in static and "transformed_images" in batch
        for k, v in batch.items():
            if (stacked and k in ("transformed_image1", "transformed_image2")) or \
                    (k == "transformed_images" and k not in static):
                continue   # halves of the stacked tensor travel with it / the halves are copied instead
            static[k].copy_(v, non_blocking=True)

    # ---- whole-step hipGraph (single process): forward + backward + fused optimiser captured once and
    # replayed per step -- for launch-bound configurations (bf16 backbones: ~35 ms of kernels per 40 ms
    # step).  Per-step scalars reach the captured optimiser kernel through device memory
    # (`LARSAdam.prepare_step`); gradients are allocated INSIDE the capture (pre-existing .grad views
    # into all-reduce buckets do not replay correctly, tools/exp/graph_capture_bisect.py) and the
    # optimiser's device-side pointer table is patched to their addresses afterwards.
    # Call it BEFORE the process runs the step eagerly on the default stream (or run everything under one
    # `torch.cuda.stream(side)` like bench.py): on this ROCm build hipStreamEndCapture crashes otherwise
    # (tools/exp/graph_capture_sizes.py: capture-first works at every size tried, eager-first never).
    def capture_step_graph(self, example_batch: Dict[str, torch.Tensor], warmup: int = 3):
        self._no_fp16_graphs()
        self._release_superseded()
        if self.world_size > 1 or self.reducer is not None:
            raise RuntimeError("capture_step_graph is single-process only (no gradient buckets)")
        if self.accumulate_grad_batches != 1:
            raise RuntimeError("capture_step_graph needs accumulate_grad_batches=1")
        self._static_batch = self._clone_batch(example_batch)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(warmup):
                self.training_micro_step(self._static_batch, i)
            # one eager forward/backward so every parameter that trains has a gradient: build the
            # optimiser's work list and stage the scalars of the step the graph will perform first
            with self._autocast():
                eager_out = self.model.training_step(self._static_batch, 0)
            self._scaled(eager_out["loss"]).backward()
            self._join_wgrad()
            self._capture_eager_out = {k: v.detach().clone() for k, v in eager_out.items()}
            self.optimizer.prepare_step()
            self.optimizer.launch_only()          # performs that step eagerly (and builds the work list)
            self.scheduler.step()
            self.global_step += 1
            self.optimizer.zero_grad(set_to_none=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with self._capturing(self._graph):
            with self._autocast():
                out = self.model.training_step(self._static_batch, 0)
            self._scaled(out["loss"]).backward()
            self._join_wgrad()
            self.optimizer.launch_only(reuse_worklist=True)
        self.optimizer.repoint_worklist()         # gradients now live in the graph's private pool
        # the captured optimiser launch reads this work list's device tables: keep it alive even if a
        # later eager step makes the optimiser build a new one
        self._graph_worklist = self.optimizer._fused_cache.get("all")
        self._static_grads = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]
        self._static_out = out
        return self

    def replay_step(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        if batch is not None and batch is not self._static_batch:
            self._load_static(self._static_batch, batch)
        self.optimizer.prepare_step()
        self._graph.replay()
        self.scheduler.step()
        self.global_step += 1
        return self._static_out

    # ---- split hipGraphs (any world size): the step is cut where its collectives are.
    #   graph A      images -> encoder -> head -> alignment -> z_local            (per-rank, ~350 launches)
    #   eager        all-gather z, NT-Xent forward, gather lse/stats/loss, NT-Xent backward -> dz_local  (~8 launches)
    #   graphs B1..  backward of graph A from dz_local, one graph per stage (head + layer4 | layer3 | the rest)
    #   eager        after each stage: its gradients -> flat buckets, SUM all-reduce launched asynchronously;
    #                after the last one: wait, fused optimiser, scheduler
    # Gradients are allocated inside the backward captures (parameters have no .grad then, see above) and
    # copied into the all-reduce buckets after each replay: one extra pass over 98 MB (~40 us) buys
    # ~800 launches per step replayed instead of issued.  Collectives stay outside the graphs, so this
    # does not depend on RCCL's capture support.
    def capture_split_graphs(self, example_batch: Dict[str, torch.Tensor], warmup: int = 3, two_stage: Optional[bool] = None):
        """two_stage (default: when the encoder is the in-tree ResNet wrapper; the name is round 2's first version --
        there are three stages now): the backward is captured as one graph per stage, cut at layer4's and layer3's
        inputs -- B1 = head + layer4, B2 = layer3, B3 = layer2..stem.  After each stage its gradients are copied into
        their buckets and all-reduced ASYNCHRONOUSLY while the next stage replays (RN-50: 65 MB, then 28 MB, travel
        under the remaining backward), so only the last stage's ~6 MB stay exposed."""
        self._no_fp16_graphs()
        self._release_superseded()
        if self.reducer is None:
            raise RuntimeError("capture_split_graphs works on the flat gradient buckets: Trainer(grad_buckets=True) "
                               "or world_size > 1")
        if self.sync_batchnorm and self.world_size > 1:
            raise RuntimeError("capture_split_graphs: synchronised BatchNorm puts collectives inside the graphs")
        model = self.model
        can_cut = self._backward_stage_of(model) is not None
        if two_stage is None:
            two_stage = can_cut
        if two_stage and not can_cut:
            raise RuntimeError("a staged backward needs the in-tree encoder (peclr_amd.encoder.ResNetModel)")
        self._static_batch = self._clone_batch(example_batch)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        if warmup < 1:
            raise ValueError("capture_split_graphs needs at least one eager step first (kernel JIT, optimiser state)")
        kacc = self.accumulate_grad_batches
        with torch.cuda.stream(side):
            for i in range(-(-warmup // kacc) * kacc):        # whole accumulation windows: ends on an optimiser step
                eager_out = self.training_micro_step(self._static_batch, i)
            self._capture_eager_out = {k: v.clone() for k, v in eager_out.items()}
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._split_count = 0
        params = [p for p in model.parameters() if p.requires_grad]
        for p in params:
            p.grad = None                     # the backward graphs allocate the gradients in the graphs' pool
        self.reducer._armed = False           # hooks stay inert: no collective inside a capture
        pool = torch.cuda.graph_pool_handle()
        self._graph_a = torch.cuda.CUDAGraph()
        seams = {}                            # features index -> (tensor that ends the later graph, leaf that starts it)

        def cut(at, t):
            seams[at] = (t, t.detach().requires_grad_())
            return seams[at][1]

        with self._capturing(self._graph_a, pool=pool, capture_error_mode="thread_local"):
            with self._autocast():
                z, row_stats, n_pairs = model._project(self._static_batch, cut=cut) if two_stage else \
                    model._project(self._static_batch)
        self._split_z, self._split_rows, self._split_n = z, row_stats, n_pairs
        self._split_dz = torch.zeros_like(z)
        # one backward graph per stage, from the loss side down: (z, dz), then (seam output, the gradient its leaf
        # received from the graph before) for every seam in descending order
        roots = [(z, self._split_dz)] + [None] * len(seams)
        self._graph_bs, per_stage, seen = [], [], set()
        for k in range(len(roots)):
            if k > 0:
                out, leaf = seams[sorted(seams, reverse=True)[k - 1]]
                roots[k] = (out, leaf.grad)
            g = torch.cuda.CUDAGraph()
            with self._capturing(g, pool=pool, capture_error_mode="thread_local"):
                torch.autograd.backward((roots[k][0],), (roots[k][1],))
                self._join_wgrad()
            self._graph_bs.append(g)
            fresh = [(p, p.grad) for p in params if p.grad is not None and id(p) not in seen]
            seen.update(id(p) for p, _ in fresh)
            per_stage.append(fresh)
        self._split_seams = seams             # keeps the seam tensors (graph-pool memory) referenced
        self._graph_b = self._graph_bs[0]
        self._graph_b2 = self._graph_bs[1] if len(self._graph_bs) > 1 else None
        self.reducer.zero_grad()              # .grad = bucket views again (optimiser + all-reduce read those)
        self._split_stages = []               # per stage: (captured gradients, their bucket views, the stage's buckets)
        for stage, pairs in enumerate(per_stage):
            buckets = [b for b in self.reducer.buckets if b.stage == stage] if two_stage else list(self.reducer.buckets)
            owned = {id(p) for b in buckets for p in b.params}
            if any(id(p) not in owned for p, _ in pairs):
                raise RuntimeError("staged backward: a gradient of one stage lives in another stage's bucket")
            self._split_stages.append(([g for _, g in pairs], [p.grad for p, _ in pairs], buckets))
        return self

    def replay_split(self, batch: Optional[Dict[str, torch.Tensor]] = None, batch_idx: Optional[int] = None,
                     is_final_batch: bool = False) -> Dict[str, torch.Tensor]:
        """One (micro-)batch through the split graphs.  accumulate_grad_batches = k > 1: every micro-batch has its
        own NT-Xent over the gathered embeddings (negatives are not pooled across micro-batches, as in the
        reference's Lightning loop); its gradients are ADDED, scaled by 1/k, into the buckets, and only the
        window's last micro-batch launches the all-reduces and steps the optimiser (window end by `batch_idx` when
        given -- fit: Lightning's rule incl. the epoch's final batch -- else by the number of replays)."""
        model = self.model
        k = self.accumulate_grad_batches
        if batch is not None and batch is not self._static_batch:
            self._load_static(self._static_batch, batch)
        self._split_count += 1
        last = (self._split_count % k == 0) if batch_idx is None else ((batch_idx + 1) % k == 0 or is_final_batch)
        self._graph_a.replay()
        z = self._split_z.detach().requires_grad_()
        loss = model._contrast(z, self._split_n, self._split_rows)     # collectives live here
        (dz,) = torch.autograd.grad(self._scaled(loss), z)
        self._split_dz.copy_(dz)
        handles = []
        for graph, (src, dst, buckets) in zip(self._graph_bs, self._split_stages):
            graph.replay()
            if k == 1:
                torch._foreach_copy_(dst, src)
            else:
                torch._foreach_add_(dst, src, alpha=1.0 / k)   # the buckets were zeroed after the last optimiser step
            if last:
                handles += self.reducer.launch(buckets)        # travels while the next stage's graph replays
        if last:
            for h in handles:
                h.wait()
            self.optimizer.step()
            self.scheduler.step()
            self.global_step += 1
            self._split_count = 0
            if k > 1:
                self.reducer.zero_grad()
        out = model._step_outputs(self._static_batch, loss)
        return {key: v.detach() for key, v in out.items()}

    # ---- gradient accumulation (accumulate_grad_batches = k > 1, single process): ONE graph of a micro-batch's
    # forward + backward.  The captured backward writes each gradient into the graph's own buffer (parameters
    # have no .grad at capture), so a replay OVERWRITES; after each replay the buffers are added, scaled by
    # 1/k, into accumulators that serve as .grad (one multi-tensor pass over ~100 MB), and every k-th replay is
    # followed by the eager fused optimiser step on the accumulators.  Negatives are per micro-batch, as in
    # the reference's Lightning loop.
    def capture_micro_graph(self, example_batch: Dict[str, torch.Tensor], warmup_windows: int = 1):
        self._no_fp16_graphs()
        self._release_superseded()
        k = self.accumulate_grad_batches
        if self.world_size > 1 or self.reducer is not None:
            raise RuntimeError("capture_micro_graph is single-process only (no gradient buckets)")
        self._static_batch = self._clone_batch(example_batch)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(max(1, warmup_windows) * k):      # whole windows: ends on an optimiser step
                self.training_micro_step(self._static_batch, i)
            self.optimizer.zero_grad(set_to_none=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with self._capturing(self._graph):
            with self._autocast():
                out = self.model.training_step(self._static_batch, 0)
            self._scaled(out["loss"]).backward()
            self._join_wgrad()
        pairs = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]
        self._micro_src = [g for _, g in pairs]
        self._micro_acc = [torch.zeros_like(g) for g in self._micro_src]
        for (p, _), a in zip(pairs, self._micro_acc):
            p.grad = a
        self._micro_count = 0
        self._static_out = out
        return self

    def replay_micro(self, batch: Optional[Dict[str, torch.Tensor]] = None, batch_idx: Optional[int] = None,
                     is_final_batch: bool = False) -> Dict[str, torch.Tensor]:
        """One micro-batch: replay, add its gradients / k into the accumulators, and step the optimiser when the
        window is full -- counted by `batch_idx` when given (fit: Lightning's rule, incl. the epoch's final batch),
        by the number of replays otherwise (bench)."""
        k = self.accumulate_grad_batches
        if batch is not None and batch is not self._static_batch:
            self._load_static(self._static_batch, batch)
        self._graph.replay()
        torch._foreach_add_(self._micro_acc, self._micro_src, alpha=1.0 / k)
        self._micro_count += 1
        last = (self._micro_count % k == 0) if batch_idx is None else ((batch_idx + 1) % k == 0 or is_final_batch)
        if last:
            self.optimizer.step()
            torch._foreach_zero_(self._micro_acc)
            self.scheduler.step()
            self.global_step += 1
            self._micro_count = 0
        return self._static_out

    def _capture_micro_in_fit(self, batch, batch_idx, is_final_batch):
        """fit() with accumulation: the first batch is trained on ONCE, eagerly; its gradients become the initial
        content of the accumulators; the micro-batch graph is then captured (a capture records, it does not run)."""
        out = self.training_micro_step(batch, batch_idx, is_final_batch)
        params = [p for p in self.model.parameters() if p.requires_grad]
        carried = {id(p): p.grad for p in params if p.grad is not None}  # empty if that batch closed a window
        for p in params:
            p.grad = None
        self._static_batch = self._clone_batch(batch)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with self._capturing(self._graph):
            with self._autocast():
                g_out = self.model.training_step(self._static_batch, 0)
            self._scaled(g_out["loss"]).backward()
            self._join_wgrad()
        pairs = [(p, p.grad) for p in params if p.grad is not None]
        self._micro_src = [g for _, g in pairs]
        self._micro_acc = [carried[id(p)] if id(p) in carried else torch.zeros_like(g) for p, g in pairs]
        for (p, _), a in zip(pairs, self._micro_acc):
            p.grad = a
        self._micro_pairs = [(p, a) for (p, _), a in zip(pairs, self._micro_acc)]
        self._micro_count = 0
        self._static_out = g_out
        return out

    def _graph_step(self, batch: Dict[str, torch.Tensor], batch_idx: int,
                    is_final_batch: bool = False) -> Dict[str, torch.Tensor]:
        """fit()'s step when hip_graph is on: capture on the first batch (which is trained on exactly
        once, by the eager step the capture routine runs), replay for equal shapes, eager otherwise."""
        sig = tuple((k, tuple(v.shape), v.dtype) for k, v in batch.items())
        if self.accumulate_grad_batches > 1:       # single process: one graph per micro-batch + accumulators
            if self._graph_sig is None:
                self._graph_sig = sig
                return self._capture_micro_in_fit(batch, batch_idx, is_final_batch)
            if sig != self._graph_sig:             # ragged batch: eager, accumulating into the same buffers
                out = self.training_micro_step(batch, batch_idx, is_final_batch)
                for p, a in self._micro_pairs:     # an optimiser step inside dropped .grad: hand the buffers back
                    if p.grad is None:
                        a.zero_()
                        p.grad = a
                return out
            out = self.replay_micro(batch, batch_idx, is_final_batch)
            return {k: v.detach().clone() for k, v in out.items()}
        if self._graph_sig is None:
            self._graph_sig = sig
            if self.reducer is None:
                self.capture_step_graph(batch, warmup=0)
            else:
                self.capture_split_graphs(batch, warmup=1)
            return self._capture_eager_out
        if not self._ranks_agree_on_replay(sig == self._graph_sig, is_final_batch):
            # off-shape (ragged last) batch on some rank: eager step on all of them.  The gradients of the previous replay are still
            # in place (a captured backward OVERWRITES its buffers, nothing zeroes them), so the eager
            # backward must not accumulate onto them.
            if self.reducer is None:
                for p, _ in self._static_grads:
                    p.grad = None
            else:
                self.reducer.zero_grad()
            out = self.training_micro_step(batch, batch_idx, is_final_batch)
            if self.reducer is None:          # hand the graph's buffers back to the parameters
                for p, g in self._static_grads:
                    p.grad = g
            return out
        out = self.replay_step(batch) if self.reducer is None else self.replay_split(batch)
        return {k: v.detach().clone() for k, v in out.items()}   # the graph's outputs are static buffers

    def fit(self, model, train_batches: Callable[[int], Iterable[Dict[str, torch.Tensor]]],
            val_batches: Optional[Callable[[int], Iterable[Dict[str, torch.Tensor]]]] = None):
        """`train_batches(epoch)` yields batch dicts already on the model's device."""
        if self.model is not model:
            self.attach(model)
        self.zero_grad()
        use_graph = (self.hip_graph and not (self.sync_batchnorm and self.world_size > 1)
                     and (self.precision != "fp16" or self._device_scaler())
                     and (self.accumulate_grad_batches == 1 or (self.world_size == 1 and self.reducer is None)))
        step = self._graph_step if use_graph else self.training_micro_step
        # with graphs the whole loop lives on one side stream: a backward on the default stream before the
        # capture would pull the legacy stream into it (see capture_step_graph)
        stream_ctx = contextlib.nullcontext()
        if use_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            stream_ctx = torch.cuda.stream(side)
        with stream_ctx:
            for epoch in range(getattr(self, "_start_epoch", 0), self.max_epochs):
                self.current_epoch = epoch
                model.train()
                outputs = [step(b, i, final) for i, (b, final) in enumerate(_flag_last(train_batches(epoch)))]
                if val_batches is not None:
                    model.eval()
                    with torch.no_grad():
                        vouts = [model.validation_step(b, i) for i, b in enumerate(val_batches(epoch))]
                    if vouts:
                        model.validation_epoch_end(vouts)
                if outputs:
                    model.training_epoch_end(outputs)
                    self._checkpoint(model, epoch)
        if use_graph:
            torch.cuda.current_stream().wait_stream(side)
        return model

    def resume(self, path: str):
        """Continue a run from a checkpoint written by `_checkpoint` / `save_checkpoint` (what Lightning's
        `resume_from_checkpoint` restores): weights and buffers, optimiser moments and step counts,
        scheduler position, epoch and global step.  Call after `attach(model)`; `fit` then starts at the
        epoch after the saved one."""
        ckpt = torch.load(path, map_location="cpu")
        self.model.load_state_dict(ckpt["state_dict"])
        if "optimizer_states" in ckpt:
            self.optimizer.load_state_dict(ckpt["optimizer_states"][0])
        if "lr_schedulers" in ckpt:
            self.scheduler.load_state_dict(ckpt["lr_schedulers"][0])
        if "native_amp_scaling_state" in ckpt and self._grad_scaler() is not None:
            self._scaler.load_state_dict(ckpt["native_amp_scaling_state"])
        self.global_step = int(ckpt.get("global_step", 0))
        self.current_epoch = int(ckpt.get("epoch", -1)) + 1
        self._start_epoch = self.current_epoch
        return self

    def _checkpoint(self, model, epoch):
        if self.checkpoint_dir is None or pdist.rank(self.process_group) != 0:
            return
        monitor = float(model.logged["checkpoint_saving_loss"]) if hasattr(model, "logged") else float("nan")
        path = os.path.join(self.checkpoint_dir, f"epoch={epoch}.ckpt")
        save_checkpoint(path, model, self.optimizer, self.scheduler, epoch, self.global_step, monitor, self._scaler)
        self._saved.append((monitor, path))
        self._saved.sort(key=lambda t: t[0])
        while self.save_top_k > 0 and len(self._saved) > self.save_top_k:
            _, worst = self._saved.pop()
            if os.path.exists(worst):
                os.remove(worst)
