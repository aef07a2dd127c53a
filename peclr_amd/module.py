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


class BaseModel(_Base):
    def __init__(self, config):
        super().__init__()
        if "resnet_size" in config.keys():
            self.encoder = get_wrapper_model(config, pretrained=config.get("pretrained", True))
        self.config = config
        self.train_metrics_epoch = {}
        self.train_metrics = {}
        self.validation_metrics_epoch = {}
        self.plot_params = {}
        self.process_group = None  # data-parallel group (None = default group / single process)

    def exclude_from_wt_decay(self, named_params: Iterator[Tuple[str, Tensor]], weight_decay: float,
                              skip_list: List[str] = ["bias", "bn"]) -> List[Dict[str, Union[list, float]]]:
        """base_model.py:30-51 -- substring match on the parameter name, quirks included: stem BN
        (`encoder.features.1.weight`), `downsample.1.weight` and `projection_head.1.weight` stay in
        the decayed group."""
        params, excluded_params = [], []
        for name, param in named_params:
            if not param.requires_grad:
                continue
            elif any(layer_name in name for layer_name in skip_list):
                excluded_params.append(param)
            else:
                params.append(param)
        return [{"params": params, "weight_decay": weight_decay},
                {"params": excluded_params, "weight_decay": 0.0}]

    def setup(self, stage: str):
        global_batch_size = self.trainer.world_size * self.config.batch_size
        self.train_iters_per_epoch = self.config.num_samples // global_batch_size

    def configure_optimizers(self) -> Tuple[list, list]:
        parameters = self.exclude_from_wt_decay(self.named_parameters(),
                                                weight_decay=self.config.opt_weight_decay)
        lr = self.config.lr * math.sqrt(self.config.batch_size * self.config.num_of_mini_batch)
        warmup_epochs = self.config.warmup_epochs * self.train_iters_per_epoch // self.config.num_of_mini_batch
        if "lr_max_epochs" in self.config.keys() and self.config["lr_max_epochs"] is not None:
            max_epochs = self.config["lr_max_epochs"] * self.train_iters_per_epoch // self.config.num_of_mini_batch
        else:
            max_epochs = self.trainer.max_epochs * self.train_iters_per_epoch // self.config.num_of_mini_batch
        if self.config.optimizer == "LARS":
            # Adam -> LARSWrapper of the reference, fused (base_model.py:62-66,90-98)
            optimizer = LARSAdam(parameters, lr=lr, lars=True)
            scheduler = LinearWarmupCosineAnnealingLR(optimizer, warmup_epochs=warmup_epochs,
                                                      max_epochs=max_epochs, warmup_start_lr=0, eta_min=0)
        else:
            optimizer = LARSAdam(parameters, lr=lr, lars=False)  # == torch.optim.Adam
            scheduler = CosineAnnealingLR(optimizer, T_max=max_epochs)
        scheduler = {"scheduler": scheduler, "interval": "step", "frequency": 1}
        return [optimizer], [scheduler]

    def training_epoch_end(self, outputs: List[dict]):
        metric_keys = outputs[0].keys()
        self.train_metrics_epoch = {key: torch.stack([x[key] for x in outputs]).mean() for key in metric_keys}
        if "loss_3d" in metric_keys:
            self.log("checkpoint_saving_loss", self.train_metrics_epoch["loss_3d"])
        else:
            self.log("checkpoint_saving_loss", self.train_metrics_epoch["loss"])

    def validation_epoch_end(self, outputs: List[dict]):
        metric_keys = outputs[0].keys()
        self.validation_metrics_epoch = {key: torch.stack([x[key] for x in outputs]).mean()
                                         for key in metric_keys}


class SimCLR(BaseModel):
    def __init__(self, config):
        super().__init__(config)
        self.projection_head = self.get_projection_head()

    def get_projection_head(self) -> nn.Sequential:
        return nn.Sequential(
            nn.Linear(self.config.projection_head_input_dim, self.config.projection_head_hidden_dim, bias=True),
            nn.BatchNorm1d(self.config.projection_head_hidden_dim),
            nn.ReLU(),
            nn.Linear(self.config.projection_head_hidden_dim, self.config.output_dim, bias=False),
        )

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

    def _project(self, batch: Dict[str, Tensor]):
        """Per-rank part of the step (no collective): images -> unit embeddings z [2N,128] (+ per-row
        projection statistics, empty here).  simclr_model.py:37-47: one F.normalize, no alignment."""
        batch_size = batch["transformed_image1"].size()[0]
        concat_batch = torch.cat((batch["transformed_image1"], batch["transformed_image2"]), dim=0)
        concat_encoding = self.get_encodings(concat_batch)
        z, row_stats = self._head_align(concat_encoding,
                                        ops.AlignSpec(n_pairs=batch_size, single_norm=True, want_stats=False))
        return z, row_stats, batch_size

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

    def _step_outputs(self, batch: dict, loss: Tensor) -> Dict[str, Tensor]:
        self.train_metrics = {**self.train_metrics, **{"loss": loss}}
        self.plot_params = {"image1": batch["transformed_image1"], "image2": batch["transformed_image2"],
                            "params": {k: v for k, v in batch.items() if "image" not in k}}
        return self.train_metrics

    def get_encodings(self, batch_images: Tensor) -> Tensor:
        return self.encoder(batch_images)

    def forward(self, x: Tensor) -> Dict[str, Tensor]:
        """simclr_model.py:54-57 (inference surface; the reference runs the encoder twice, the two
        results are identical in eval mode, so it is run once here)."""
        embedding = self.encoder(x)
        projection = self.projection_head(embedding)  # stock torch modules: not on the training path
        return {"embedding": embedding, "projection": projection}

    def training_step(self, batch: dict, batch_idx: int) -> Dict[str, Tensor]:
        return self._step_outputs(batch, self.contrastive_step(batch))

    def validation_step(self, batch: dict, batch_idx: int) -> Dict[str, Tensor]:
        loss = self.contrastive_step(batch)
        self.plot_params = {"image1": batch["transformed_image1"], "image2": batch["transformed_image2"],
                            "params": {k: v for k, v in batch.items() if "image" not in k}}
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

    def _project(self, batch: Dict[str, Tensor]):
        batch_transform = torch.cat((batch["transformed_image1"], batch["transformed_image2"]), dim=0)
        encodings = self.encoder(batch_transform)
        spec = self._spec(batch)
        z, row_stats = self._head_align(encodings, spec)
        return z, row_stats, spec.n_pairs

    def get_transformed_projections(self, batch: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
        z, row_stats, n = self._project(batch)
        stats16 = row_stats.view(2, n, 8).mean(dim=1).reshape(16)
        self.train_metrics = {**self.train_metrics, **dict(zip(STAT_KEYS, stats16.unbind()))}
        return z[:n], z[n:]

    def get_projection_stats(self, projection: Tensor, name: str) -> dict:
        """hybrid2_model.py:92-106, kept for callers; the training path gets these from the kernel."""
        projection_mean = torch.mean(projection, dim=1)
        projection_median = torch.median(projection, dim=1).values
        projection_min = torch.min(projection, dim=1).values
        projection_max = torch.max(projection, dim=1).values
        out = {}
        for c, a in enumerate("xy"):
            out[f"{name}{a}_mean"] = torch.mean(projection_mean, dim=0)[c]
            out[f"{name}{a}_median"] = torch.mean(projection_median, dim=0)[c]
            out[f"{name}{a}_min"] = torch.mean(projection_min, dim=0)[c]
            out[f"{name}{a}_max"] = torch.mean(projection_max, dim=0)[c]
        return out


def get_model(experiment_type: str):
    """experiments/utils.py:564-584 for the two model classes that exist on this path."""
    if experiment_type == "simclr":
        return SimCLR
    if experiment_type == "hybrid2":
        return Hybrid2Model
    raise NotImplementedError(f"experiment type {experiment_type!r} is outside the PeCLR pretraining path")
