"""The reference's module surface on the MI355X kernels: BaseModel -> SimCLR -> Hybrid2Model
(base_model.py:13-127, simclr_model.py:10-76, hybrid2_model.py:16-106).

Same constructor (`config`), same hooks (`setup`, `configure_optimizers`, `training_step`,
`validation_step`, `training_epoch_end`, `validation_epoch_end`, `forward`, `get_encodings`,
`contrastive_step`, `get_transformed_projections`), same attributes (`encoder`, `projection_head`,
`config`, `train_metrics`, `train_metrics_epoch`, `validation_metrics_epoch`, `plot_params`), same
state_dict keys -- so it drops into `src/experiments/peclr_training.py` in place of
`get_model("hybrid2")` (experiments/utils.py:570-574).  It subclasses LightningModule when
pytorch_lightning is importable and a plain nn.Module otherwise (`peclr_amd.trainer.Trainer`
drives it then).

What changed underneath: the ~100 stock-op launches and the CPU round trip of
`get_transformed_projections` + `vanila_contrastive_loss` are six HIP launches forward and eight
backward (`peclr_amd.ops`); `projection_head` is still an nn.Sequential of the same four modules
and only holds the parameters/buffers -- its arithmetic runs in the kernels.
"""
from __future__ import annotations

import math
from typing import Dict, Iterator, List, Tuple, Union

import torch
from torch import Tensor, nn
from torch.optim.lr_scheduler import CosineAnnealingLR

from . import ops
from .encoder import get_wrapper_model
from .optim import LARSAdam, LinearWarmupCosineAnnealingLR

try:  # optional: the real base class when the caller's environment has it
    from pytorch_lightning.core.lightning import LightningModule as _Base  # type: ignore
except Exception:  # pragma: no cover - not installed on the target image
    try:
        from pytorch_lightning import LightningModule as _Base  # type: ignore
    except Exception:

        class _Base(nn.Module):
            """Duck-typed stand-in: what the hot path uses of LightningModule."""

            def __init__(self):
                super().__init__()
                self.trainer = None
                self.logged: Dict[str, Tensor] = {}

            def log(self, name, value, *args, **kwargs):
                self.logged[name] = value


STAT_KEYS = tuple(f"{n}{a}_{s}" for n in ("proj1", "proj2") for a in ("x", "y")
                  for s in ("mean", "median", "min", "max"))
TEMPERATURE = 0.5  # vanila_contrastive_loss default; not a config value (utils.py:154)
_NO_DECAY_MARKERS = ("bias", "bn")  # base_model.py:34


def _epoch_mean(step_dicts: List[dict]) -> Dict[str, Tensor]:
    """Mean over an epoch's step dicts, key by key, in the first dict's key order (a missing key in a
    later dict is a KeyError, as in base_model.py:107-110)."""
    return {name: torch.stack([d[name] for d in step_dicts]).mean() for name in step_dicts[0]}


class BaseModel(_Base):
    """base_model.py:13-127.  `strict_reference` (config key, default False) switches on the
    reference's wasteful-but-harmless quirks that this build otherwise skips (today: `forward`
    running the encoder twice)."""

    def __init__(self, config):
        super().__init__()
        if "resnet_size" in config.keys():
            self.encoder = get_wrapper_model(config, pretrained=config.get("pretrained", True))
        self.config = config
        self.train_metrics, self.train_metrics_epoch, self.validation_metrics_epoch = {}, {}, {}
        self.plot_params = {}
        self.process_group = None  # data-parallel group (None = default group / single process)

    # ---- optimiser plumbing
    def exclude_from_wt_decay(self, named_params: Iterator[Tuple[str, Tensor]], weight_decay: float,
                              skip_list=_NO_DECAY_MARKERS) -> List[Dict[str, Union[list, float]]]:
        """Two Adam parameter groups: [decayed, not decayed].  A trainable parameter is kept out of
        weight decay when its NAME contains one of `skip_list` -- a substring test, so only the
        residual blocks' `bn1/bn2/bn3` and every `.bias` match; the stem BatchNorm
        (`encoder.features.1.weight`), `downsample.1.weight` and `projection_head.1.weight` stay
        decayed (base_model.py:30-51)."""
        groups = ({"params": [], "weight_decay": weight_decay}, {"params": [], "weight_decay": 0.0})
        for name, tensor in named_params:
            if tensor.requires_grad:
                groups[any(marker in name for marker in skip_list)]["params"].append(tensor)
        return list(groups)

    def setup(self, stage: str):
        """Micro-batches per epoch over all ranks (base_model.py:53-55)."""
        self.train_iters_per_epoch = self.config.num_samples // (self.trainer.world_size * self.config.batch_size)

    def _optimizer_steps(self, epochs: int) -> int:
        """Epochs -> optimiser steps, with the reference's operator order: (epochs * iters) // accum."""
        return epochs * self.train_iters_per_epoch // self.config.num_of_mini_batch

    def configure_optimizers(self) -> Tuple[list, list]:
        """base_model.py:57-104: Adam at lr*sqrt(batch*accum) over the two groups above; "LARS" wraps
        it (pl_bolts LARSWrapper) and schedules linear warm-up + cosine per optimiser step, anything
        else is plain Adam under CosineAnnealingLR.  Both arms are ONE fused HIP optimiser here."""
        cfg = self.config
        groups = self.exclude_from_wt_decay(self.named_parameters(), weight_decay=cfg.opt_weight_decay)
        horizon = cfg["lr_max_epochs"] if cfg.get("lr_max_epochs") is not None else self.trainer.max_epochs
        total_steps = self._optimizer_steps(horizon)
        use_lars = cfg.optimizer == "LARS"
        optimizer = LARSAdam(groups, lr=cfg.lr * math.sqrt(cfg.batch_size * cfg.num_of_mini_batch), lars=use_lars,
                             write_back=bool(cfg.get("strict_reference", False)))
        if use_lars:
            schedule = LinearWarmupCosineAnnealingLR(optimizer, warmup_epochs=self._optimizer_steps(cfg.warmup_epochs),
                                                     max_epochs=total_steps, warmup_start_lr=0, eta_min=0)
        else:
            schedule = CosineAnnealingLR(optimizer, T_max=total_steps)
        return [optimizer], [{"scheduler": schedule, "interval": "step", "frequency": 1}]

    # ---- epoch-end hooks (a11).  The monitored quantity is the TRAINING epoch's mean loss -- "loss_3d" when the step
    # dict has one (the supervised subclasses of the reference inherit this hook), "loss" otherwise (base_model.py:111-115).
    def training_epoch_end(self, outputs: List[dict]):
        self.train_metrics_epoch = _epoch_mean(outputs)
        monitored = "loss_3d" if "loss_3d" in self.train_metrics_epoch else "loss"
        self.log("checkpoint_saving_loss", self.train_metrics_epoch[monitored])

    def validation_epoch_end(self, outputs: List[dict]):
        self.validation_metrics_epoch = _epoch_mean(outputs)


class SimCLR(BaseModel):
    def __init__(self, config):
        super().__init__(config)
        self.projection_head = self.get_projection_head()

    def get_projection_head(self) -> nn.Sequential:
        """Linear(bias) -> BatchNorm1d -> ReLU -> Linear(no bias) (simclr_model.py:20-35).  The
        Sequential only OWNS the parameters and buffers (state_dict keys projection_head.{0,1,3}.*);
        the training path's arithmetic runs in the HIP kernels."""
        d_in, d_hid, d_out = (self.config[k] for k in ("projection_head_input_dim", "projection_head_hidden_dim",
                                                       "output_dim"))
        return nn.Sequential(nn.Linear(d_in, d_hid), nn.BatchNorm1d(d_hid), nn.ReLU(),
                             nn.Linear(d_hid, d_out, bias=False))

    # ---- kernels
    def _head_align(self, encodings: Tensor, spec: ops.AlignSpec) -> Tuple[Tensor, Tensor]:
        lin1, bn, _, lin2 = self.projection_head
        state = ops.BNState(training=bn.training or not bn.track_running_stats, eps=bn.eps,
                            momentum=bn.momentum if bn.momentum is not None else 0.1,
                            running_mean=bn.running_mean, running_var=bn.running_var,
                            num_batches_tracked=bn.num_batches_tracked,
                            sync_group=getattr(self, "sync_bn_group", None))
        return ops.head_align(encodings, lin1.weight, lin1.bias, bn.weight, bn.bias, lin2.weight, state, spec)

    def _loss(self, z: Tensor, n_pairs: int, row_stats=None):
        loss, stats16, _ = ops.ntxent(z, n_pairs, TEMPERATURE, row_stats, self.process_group)
        return loss, stats16

    @staticmethod
    def _two_views(batch: Dict[str, Tensor]) -> Tensor:
        """Both views as one [2N, 3, H, W] tensor, view 1 first (hybrid2_model.py:30-32).  A batch that
        already carries them stacked (`transformed_images`: TwoViewAugmenter and bench.py emit the two
        views as halves of one buffer) is used as is: no copy."""
        stacked = batch.get("transformed_images")
        if stacked is not None:
            return stacked
        return torch.cat((batch["transformed_image1"], batch["transformed_image2"]), dim=0)

    def _project(self, batch: Dict[str, Tensor], cut=None):
        """Per-rank part of the step (no collective): images -> unit embeddings z [2N,128] (+ per-row
        projection statistics, empty here).  simclr_model.py:37-47: one F.normalize, no alignment."""
        n_pairs = batch["transformed_image1"].size(0)
        views = self._two_views(batch)
        z, row_stats = self._head_align(self.get_encodings(views) if cut is None else self.encoder(views, cut=cut),
                                        ops.AlignSpec(n_pairs=n_pairs, single_norm=True, want_stats=False))
        return z, row_stats, n_pairs

    def _contrast(self, z: Tensor, n_pairs: int, row_stats: Tensor) -> Tensor:
        """Cross-rank part of the step: NT-Xent over the gathered embeddings (+ the batch means of the
        projection statistics, which ride in the loss finalize launch)."""
        if row_stats is None or row_stats.numel() == 0:
            return self._loss(z, n_pairs)[0]
        loss, stats16 = self._loss(z, n_pairs, row_stats)
        self.train_metrics = {**self.train_metrics, **dict(zip(STAT_KEYS, stats16.unbind()))}
        return loss

    def contrastive_step(self, batch: Dict[str, Tensor]) -> Tensor:
        z, row_stats, n = self._project(batch)
        return self._contrast(z, n, row_stats)

    def _remember_views(self, batch: dict):
        """What the reference's Comet callback plots (upload_comet_logs.py:91-98): the two image
        tensors and every non-image batch entry."""
        extras = {name: value for name, value in batch.items() if "image" not in name}
        self.plot_params = {"image1": batch["transformed_image1"], "image2": batch["transformed_image2"],
                            "params": extras}

    def _step_outputs(self, batch: dict, loss: Tensor) -> Dict[str, Tensor]:
        self.train_metrics = {**self.train_metrics, "loss": loss}
        self._remember_views(batch)
        return self.train_metrics

    def get_encodings(self, batch_images: Tensor) -> Tensor:
        return self.encoder(batch_images)

    def forward(self, x: Tensor) -> Dict[str, Tensor]:
        """Inference surface (simclr_model.py:54-57): encoder features and their projection through the
        stock torch head modules (not the training path).  The reference evaluates the encoder a
        second time for the "embedding" entry; in eval mode both results are identical, so that is
        only done with `config.strict_reference` (in train mode the second pass also updates the
        BatchNorm running statistics once more, which is what the flag reproduces)."""
        embedding = self.encoder(x)
        out = {"embedding": embedding, "projection": self.projection_head(embedding)}
        if self.config.get("strict_reference", False):
            out["embedding"] = self.encoder(x)
        return out

    def training_step(self, batch: dict, batch_idx: int) -> Dict[str, Tensor]:
        """Returns `self.train_metrics` itself: projection statistics (if any) then "loss", which is the
        entry Lightning back-propagates (simclr_model.py:59-67)."""
        return self._step_outputs(batch, self.contrastive_step(batch))

    def validation_step(self, batch: dict, batch_idx: int) -> Dict[str, Tensor]:
        loss = self.contrastive_step(batch)
        self._remember_views(batch)
        return {"loss": loss}


class Hybrid2Model(SimCLR):
    """PeCLR: equivariance is preserved by transforming the projection space (hybrid2_model.py:16-106)."""

    def __init__(self, config):
        super().__init__(config)

    def _spec(self, batch: Dict[str, Tensor]) -> ops.AlignSpec:
        image1_shape = batch["transformed_image1"].size()[-2:]
        image2_shape = batch["transformed_image2"].size()[-2:]
        if tuple(image1_shape) != tuple(image2_shape):
            raise ValueError("the two views must have the same spatial size (they are concatenated, "
                             "hybrid2_model.py:30-32)")
        n_pairs = batch["transformed_image1"].size(0)
        crop = "crop" in self.config.augmentation
        rotate = "rotate" in self.config.augmentation
        spec = ops.AlignSpec(n_pairs=n_pairs, crop=crop, rotate=rotate)
        if crop:  # x over shape[0], y over shape[1] -- the reference's quirk (hybrid2_model.py:59-73)
            spec.jitter = tuple(batch[k].contiguous() for k in ("jitter_x_1", "jitter_x_2", "jitter_y_1",
                                                                "jitter_y_2"))
            spec.extents = (float(image1_shape[0]), float(image1_shape[1]))
        if rotate:
            spec.angles = (batch["angle_1"].contiguous(), batch["angle_2"].contiguous())
        return spec

    def _project(self, batch: Dict[str, Tensor], cut=None):
        """cut: see `peclr_amd.encoder.ResNetModel.forward` (only the in-tree encoder takes it)."""
        views = self._two_views(batch)
        encodings = self.encoder(views) if cut is None else self.encoder(views, cut=cut)
        spec = self._spec(batch)
        z, row_stats = self._head_align(encodings, spec)
        return z, row_stats, spec.n_pairs

    def get_transformed_projections(self, batch: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
        z, row_stats, n = self._project(batch)
        stats16 = row_stats.view(2, n, 8).mean(dim=1).reshape(16)
        self.train_metrics = {**self.train_metrics, **dict(zip(STAT_KEYS, stats16.unbind()))}
        return z[:n], z[n:]

    def get_projection_stats(self, projection: Tensor, name: str) -> dict:
        """Torch spelling of the 8 per-view statistics for external callers (hybrid2_model.py:92-106):
        per-sample mean / lower median / min / max over the 64 points of `projection` [N,64,2], averaged
        over the batch, split into x and y.  The training path gets the same numbers from the align
        kernel (key order: STAT_KEYS)."""
        per_sample = {"mean": projection.mean(dim=1), "median": projection.median(dim=1).values,
                      "min": projection.amin(dim=1), "max": projection.amax(dim=1)}
        batch_mean = {stat: v.mean(dim=0) for stat, v in per_sample.items()}
        return {f"{name}{axis}_{stat}": batch_mean[stat][c] for c, axis in enumerate("xy") for stat in batch_mean}


def get_model(experiment_type: str):
    """experiments/utils.py:564-584 for the two model classes that exist on this path."""
    if experiment_type == "simclr":
        return SimCLR
    if experiment_type == "hybrid2":
        return Hybrid2Model
    raise NotImplementedError(f"experiment type {experiment_type!r} is outside the PeCLR pretraining path")
