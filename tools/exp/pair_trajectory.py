#!/usr/bin/env python
"""Loss trajectories of the same fp32 run (same init, a fresh seeded batch every step) in three arithmetic arms:
  pair      forward / input-gradient GEMMs as three fp16 products of scaled hi + lo operands (the default since round 6)
  six       every GEMM as six bf16 products of exact three-way splits (PECLR_X6_PAIR=0: rounds 2 - 5)
  stock     fp32 backbone on stock PyTorch / MIOpen ops (no fused glue, no in-tree convolutions)
Training is chaotic: two fp32 implementations of the same step drift apart from rounding alone.  What this measures is whether
the pair arm stays inside the spread that the six-product arm and MIOpen's fp32 kernels have between THEM.

    python tools/exp/pair_trajectory.py [--resnet 50 --pairs 64 --size 224 --steps 60] > profiles/r06_pair_trajectory.json
"""
import argparse
import copy
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, ".miopen", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(ROOT, ".miopen", "cache"))
import torch  # noqa: E402


def trajectories(resnet, pairs, size, steps, seed, n_batches):
    from bench import synthetic_batch
    from peclr_amd import Hybrid2Model, Trainer, bn2d, hybrid2_config
    from peclr_amd.bn2d import enable_hip_batchnorm

    warnings.simplefilter("ignore")
    dev = torch.device("cuda", 0)
    din = 512 if resnet in ("18", "34") else 2048
    cfg = hybrid2_config(resnet_size=resnet, projection_head_input_dim=din, augmentation=["crop", "rotate"],
                         batch_size=pairs, num_samples=pairs * n_batches, warmup_epochs=1, pretrained=False)
    torch.manual_seed(seed)
    base = Hybrid2Model(cfg).to(dev).train()
    base.encoder = base.encoder.to(memory_format=torch.channels_last)
    batches = [synthetic_batch(pairs, size, seed + 17 * i, dev, channels_last=True) for i in range(n_batches)]
    out = {}
    for arm, fused, pair in (("pair", True, True), ("six", True, False), ("stock", False, False)):
        model = copy.deepcopy(base)
        enable_hip_batchnorm(model.encoder, fused)
        with bn2d.routing(x6_pair=pair):
            tr = Trainer(max_epochs=100, precision="fp32").attach(model)
            tr.zero_grad()
            losses = []
            for i in range(steps):
                losses.append(float(tr.training_micro_step(batches[i % n_batches], i)["loss"]))    # (k = 1: steps the optimiser)
        out[arm] = losses
        del model, tr
        torch.cuda.empty_cache()
    return out


def summarise(t):
    import math

    def spread(a, b):
        d = [abs(x - y) for x, y in zip(t[a], t[b])]
        return {"max": max(d), "mean": sum(d) / len(d), "last": d[-1], "first_step": d[0]}

    return {"pair_vs_six": spread("pair", "six"), "pair_vs_stock": spread("pair", "stock"), "six_vs_stock": spread("six", "stock"),
            "first": {k: v[0] for k, v in t.items()}, "final": {k: v[-1] for k, v in t.items()},
            "drop_from_step0": {k: v[0] - v[-1] for k, v in t.items()},
            "nan": any(math.isnan(x) for v in t.values() for x in v)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--resnet", default="50")
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--seed", type=int, default=5)
    a = ap.parse_args()
    t = trajectories(a.resnet, a.pairs, a.size, a.steps, a.seed, a.batches)
    print(json.dumps({f"resnet{a.resnet}_2x{a.pairs}_{a.size}_{a.steps}steps": {"summary": summarise(t), "loss": t}}, indent=1))
