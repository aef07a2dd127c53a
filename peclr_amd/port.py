"""`peclr_to_torchvision` (port_model.py:7-48): copy the encoder of a PeCLR checkpoint into a
torchvision-layout ResNet, positionally, and the checkpoint helpers around it
(utils.py:189-225).  Same signature and behaviour, including the print-and-break on a name
mismatch and the bare `Exception` for a non-ResNet argument.
"""
from __future__ import annotations

import os
import torch

from . import resnet as _resnet

try:  # the reference checks against torchvision's class; accept it when it is installed
    import torchvision as _tv  # type: ignore

    _RESNET_TYPES = (_resnet.ResNet, _tv.models.ResNet)
except Exception:  # torchvision is not on the target image
    _RESNET_TYPES = (_resnet.ResNet,)


def peclr_to_torchvision(resnet_model, path_to_peclr_weights):
    """Load the encoder of a PeCLR checkpoint into a torchvision-layout ResNet, in place; returns None.

    Contract (port_model.py:7-48): the checkpoint's `state_dict` entries whose key contains "features"
    are matched POSITIONALLY with `resnet_model.state_dict()` (conv1, bn1, layer1..4 come first there,
    `fc.*` last and is never reached).  Before each copy the last dotted component of the two keys
    ("weight", "running_mean", ...) must agree; on a disagreement, or when a copy fails (shape
    mismatch = a different ResNet size), a message is printed and the remaining entries are left
    untouched -- no exception.  A model that is not a ResNet raises a bare `Exception`."""
    checkpoint = torch.load(path_to_peclr_weights, map_location="cpu")
    print(checkpoint.keys())
    if not isinstance(resnet_model, _RESNET_TYPES):
        raise Exception("The selected model is not of type ResNet from torch vision!")
    encoder_entries = [(k, v) for k, v in checkpoint["state_dict"].items() if "features" in k]
    destination = resnet_model.state_dict()   # name -> the live parameter / buffer storage
    for (dst_key, dst), (src_key, src) in zip(destination.items(), encoder_entries):
        if dst_key.rsplit(".", 1)[-1] != src_key.rsplit(".", 1)[-1]:
            print("PeCLR layers don't match with Resnet layer ")
            return
        try:
            dst.copy_(src)
        except Exception as exc:  # noqa: BLE001 -- the reference reports and stops, whatever the cause
            print("The models are not compatible!")
            print(f"Exception :{exc}")
            return
    if len(encoder_entries) > len(destination):  # the reference indexes past the end of the ResNet's list here
        raise IndexError("the checkpoint holds more encoder entries than the ResNet has state_dict entries")


def _checkpoint_dir(experiment_name: str) -> str:
    """`$SAVED_MODELS_BASE_PATH/<experiment_name>/checkpoints` (constants.py:5 reads the variable at
    import; here it is read per call).  An absolute `experiment_name` wins, as with os.path.join."""
    base = os.environ.get("SAVED_MODELS_BASE_PATH")
    if base is None and not os.path.isabs(experiment_name):
        raise KeyError("SAVED_MODELS_BASE_PATH is not set (the reference resolves checkpoints under "
                       "$SAVED_MODELS_BASE_PATH/<experiment>/checkpoints)")
    return os.path.join(base or "", experiment_name, "checkpoints")


def get_latest_checkpoint(experiment_name: str, checkpoint: str = "") -> str:
    """utils.py:189-206.  The named checkpoint, or the one with the largest epoch among the
    `epoch=<K>.ckpt` files of the experiment's checkpoint directory."""
    folder = _checkpoint_dir(experiment_name)
    if checkpoint == "":
        checkpoint = max(os.listdir(folder), key=lambda name: int(name[len("epoch="):-len(".ckpt")]))
    return os.path.join(folder, checkpoint)


def get_encoder_state_dict(saved_model_path: str, checkpoint: str, map_location=None) -> dict:
    """utils.py:209-225: the `encoder.*` entries of an experiment's checkpoint with the 8-character
    prefix removed, ready for `ResNetModel.load_state_dict`.  `map_location` is an addition (the
    reference loads onto the device the checkpoint was saved from)."""
    state = torch.load(get_latest_checkpoint(saved_model_path, checkpoint), map_location=map_location)["state_dict"]
    return {key[len("encoder."):]: value for key, value in state.items() if "encoder" in key}


def restore_model(model: torch.nn.Module, experiment_key: str, checkpoint: str = ""):
    """experiments/utils.py:535-546: load the newest (or the named) checkpoint of an experiment."""
    path = get_latest_checkpoint(experiment_key, checkpoint)
    print(f"Restoring {path}")
    model.load_state_dict(torch.load(path, map_location="cpu")["state_dict"])
    return model


def save_checkpoint(path: str, model: torch.nn.Module, optimizer=None, scheduler=None, epoch: int = 0,
                    global_step: int = 0, monitor: float = None, scaler=None):
    """Lightning-shaped checkpoint ({"state_dict", "epoch", "global_step", ...}) so files
    interoperate with `peclr_to_torchvision` and the published `.pth` weights (README.md:84-96)."""
    ckpt = {"state_dict": model.state_dict(), "epoch": epoch, "global_step": global_step}
    if optimizer is not None:
        ckpt["optimizer_states"] = [optimizer.state_dict()]
    if scheduler is not None:
        ckpt["lr_schedulers"] = [scheduler.state_dict()]
    if scaler is not None:  # precision=16: Lightning's key for the GradScaler state
        ckpt["native_amp_scaling_state"] = scaler.state_dict()
    if monitor is not None:
        ckpt["checkpoint_saving_loss"] = float(monitor)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)
