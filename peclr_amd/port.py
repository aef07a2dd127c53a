"""`peclr_to_torchvision` (port_model.py:7-48): copy the encoder of a PeCLR checkpoint into a
torchvision-layout ResNet, positionally, and the checkpoint helpers around it
(utils.py:189-225).  Same signature and behaviour, including the print-and-break on a name
mismatch and the bare `Exception` for a non-ResNet argument.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch

from . import resnet as _resnet

try:  # the reference checks against torchvision's class; accept it when it is installed
    import torchvision as _tv  # type: ignore

    _RESNET_TYPES = (_resnet.ResNet, _tv.models.ResNet)
except Exception:  # torchvision is not on the target image
    _RESNET_TYPES = (_resnet.ResNet,)


def peclr_to_torchvision(resnet_model, path_to_peclr_weights):
    """Copies parameters from a trained PeCLR model to a corresponding torchvision-layout ResNet.
    All the weights until the fc layer are copied (every state_dict entry whose key contains
    "features", in order); `fc` is never touched.  Mutates `resnet_model` in place, returns None."""
    peclr_weights = torch.load(path_to_peclr_weights, map_location=torch.device("cpu"))
    print(peclr_weights.keys())
    peclr_state_dict = peclr_weights["state_dict"]
    if isinstance(resnet_model, _RESNET_TYPES):
        resnet_state_dict_list = list(resnet_model.state_dict().items())
        peclr_state_dict_list = [(key, peclr_state_dict[key]) for key in peclr_state_dict if "features" in key]
        own_state = resnet_model.state_dict()
        for idx in range(len(peclr_state_dict_list)):
            if resnet_state_dict_list[idx][0].split(".")[-1] != peclr_state_dict_list[idx][0].split(".")[-1]:
                print("PeCLR layers don't match with Resnet layer ")
                break
            name = resnet_state_dict_list[idx][0]
            param = peclr_state_dict_list[idx][1]
            try:
                own_state[name].copy_(param)
            except Exception as e:
                print("The models are not compatible!")
                print(f"Exception :{e}")
                break
    else:
        raise Exception("The selected model is not of type ResNet from torch vision!")


def get_latest_checkpoint(checkpoint_dir: str, checkpoint: str = "") -> str:
    """utils.py:189-206: newest `epoch=K.ckpt` (sorted by int(name[6:-5])) unless one is named."""
    if checkpoint:
        return os.path.join(checkpoint_dir, checkpoint)
    names = sorted(os.listdir(checkpoint_dir), key=lambda x: int(x[6:-5]))
    return os.path.join(checkpoint_dir, names[-1])


def get_encoder_state_dict(saved_model_path: str) -> OrderedDict:
    """utils.py:209-225: keep `encoder.*` entries and strip the 8-character prefix."""
    saved_state_dict = torch.load(saved_model_path, map_location="cpu")["state_dict"]
    out = OrderedDict()
    for key, value in saved_state_dict.items():
        if "encoder" in key:
            out[key[8:]] = value
    return out


def restore_model(model: torch.nn.Module, checkpoint_dir: str, checkpoint: str = ""):
    """experiments/utils.py:535-546: load the newest (or the named) checkpoint's `state_dict`."""
    path = get_latest_checkpoint(checkpoint_dir, checkpoint)
    print(f"Restoring {path}")
    model.load_state_dict(torch.load(path, map_location="cpu")["state_dict"])
    return model


def save_checkpoint(path: str, model: torch.nn.Module, optimizer=None, scheduler=None, epoch: int = 0,
                    global_step: int = 0, monitor: float = None):
    """Lightning-shaped checkpoint ({"state_dict", "epoch", "global_step", ...}) so files
    interoperate with `peclr_to_torchvision` and the published `.pth` weights (README.md:84-96)."""
    ckpt = {"state_dict": model.state_dict(), "epoch": epoch, "global_step": global_step}
    if optimizer is not None:
        ckpt["optimizer_states"] = [optimizer.state_dict()]
    if scheduler is not None:
        ckpt["lr_schedulers"] = [scheduler.state_dict()]
    if monitor is not None:
        ckpt["checkpoint_saving_loss"] = float(monitor)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)
