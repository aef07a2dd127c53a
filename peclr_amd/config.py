"""Attribute-dict config, standing in for `easydict.EasyDict` (not installed here).

The reference reads its hyper-parameters as `config.key`, `config["key"]` and
`"key" in config.keys()` (base_model.py:21,60-88; simclr_model.py:23-31;
hybrid2_model.py:58,76), so any mapping with attribute access works, including a real
EasyDict when the caller has one.
"""
from __future__ import annotations

import json
from typing import Any, Mapping

# hybrid2_config.json / training_config.json values of the reference
HYBRID2_DEFAULTS = {
    "batch_size": 128, "lr": 1e-4, "opt_weight_decay": 1e-6, "output_dim": 128,
    "projection_head_hidden_dim": 512, "projection_head_input_dim": 2048, "warmup_epochs": 10,
    "num_of_mini_batch": 1, "augmentation": [], "optimizer": "LARS", "resnet_size": "50",
}


class Config(dict):
    def __init__(self, d: Mapping[str, Any] = None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, Mapping) and not isinstance(v, Config):
            v = Config(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @classmethod
    def from_json(cls, path: str) -> "Config":
        with open(path) as f:
            return cls(json.load(f))


def hybrid2_config(**overrides) -> Config:
    """The reference's hybrid2_config.json defaults, plus the three keys that
    `update_model_params` injects (experiments/utils.py:608-615)."""
    cfg = Config(HYBRID2_DEFAULTS)
    cfg.update(num_samples=overrides.pop("num_samples", 32560 + 44994))
    for k, v in overrides.items():
        cfg[k] = v
    return cfg
