#!/usr/bin/env python
"""Loss trajectories of the same run in three arithmetic arms (same init, same batch every step):
  fp32        fp32 backbone on the fused NHWC glue
  bf16_fused  bf16-autocast backbone on the fused NHWC glue (bn2d bf16 I/O, gemm_add_bf16), fp32 head
  bf16_stock  bf16-autocast backbone on stock PyTorch/MIOpen BatchNorm / add / ReLU ops, fp32 head
to separate "bf16 autocast learns slower in the warm-up" (bf16_stock vs fp32: inherent to casting fp32
master weights to bf16 every forward) from "the fused bf16 kernels are off" (bf16_fused vs bf16_stock).

    python tools/bf16_trajectory.py [--resnet 50 --pairs 128 --size 224 --steps 20] > profiles/r02_bf16_trajectory.json
"""
import argparse
import copy
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, ".miopen", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(ROOT, ".miopen", "cache"))
import torch  # noqa: E402


def trajectories(resnet="50", pairs=128, size=224, steps=20, seed=5, lr=None):
    from bench import synthetic_batch
    from peclr_amd import Hybrid2Model, Trainer, hybrid2_config
    from peclr_amd.bn2d import enable_hip_batchnorm

    warnings.simplefilter("ignore")
    dev = torch.device("cuda", 0)
    din = 512 if resnet in ("18", "34") else 2048
    over = {} if lr is None else {"lr": lr}
    cfg = hybrid2_config(resnet_size=resnet, projection_head_input_dim=din, augmentation=["crop", "rotate"],
                         batch_size=pairs, pretrained=False, **over)
    torch.manual_seed(seed)
    base = Hybrid2Model(cfg).to(dev).train()
    base.encoder = base.encoder.to(memory_format=torch.channels_last)
    batch = synthetic_batch(pairs, size, seed, dev, channels_last=True)
    out = {}
    for arm, precision, fused in (("fp32", "fp32", True), ("bf16_fused", "bf16", True), ("bf16_stock", "bf16", False)):
        model = copy.deepcopy(base)
        enable_hip_batchnorm(model.encoder, fused)
        tr = Trainer(max_epochs=100, precision=precision).attach(model)
        tr.zero_grad()
        out[arm] = [float(tr.training_micro_step(batch, i)["loss"]) for i in range(steps)]
        del model, tr
        torch.cuda.empty_cache()
    return out


def summarise(t):
    import math

    rel = [abs(a - b) / abs(b) for a, b in zip(t["bf16_fused"], t["bf16_stock"])]
    return {"max_rel_fused_vs_stock": max(rel), "final": {k: v[-1] for k, v in t.items()},
            "drop_from_step0": {k: v[0] - v[-1] for k, v in t.items()},
            "gap_bf16_stock_to_fp32_final": t["bf16_stock"][-1] - t["fp32"][-1],
            "gap_bf16_fused_to_fp32_final": t["bf16_fused"][-1] - t["fp32"][-1], "nan": any(math.isnan(x) for v in t.values() for x in v)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    res = {}
    for name, kw in (("resnet18_2x32_224", dict(resnet="18", pairs=32, size=224)),
                     ("resnet50_2x128_224", dict(resnet="50", pairs=128, size=224)),
                     # the same run with the learning rate the schedule reaches AFTER its warm-up (x240): the weight
                     # updates are then far above bf16 resolution
                     ("resnet50_2x128_224_lr_x240", dict(resnet="50", pairs=128, size=224, lr=1e-4 * 240))):
        t = trajectories(steps=a.steps, **kw)
        res[name] = {"loss": t, "summary": summarise(t)}
    print(json.dumps(res, indent=1))
