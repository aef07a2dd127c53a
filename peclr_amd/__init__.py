"""peclr_amd: MI355X-native PeCLR pretraining step (ResNet encoder + equivariant NT-Xent).

Host side of libpeclr_hip.so behind the reference's own module surface:

    from peclr_amd import Hybrid2Model, SimCLR, peclr_to_torchvision, hybrid2_config
"""
from .augment import TwoViewAugmenter
from .config import Config, hybrid2_config
from .module import BaseModel, Hybrid2Model, SimCLR, get_model
from .port import (get_encoder_state_dict, get_latest_checkpoint, peclr_to_torchvision, restore_model,
                   save_checkpoint)
from .trainer import Trainer

__all__ = ["Config", "hybrid2_config", "BaseModel", "SimCLR", "Hybrid2Model", "get_model",
           "peclr_to_torchvision", "get_encoder_state_dict", "get_latest_checkpoint", "save_checkpoint", "restore_model",
           "Trainer", "TwoViewAugmenter"]
