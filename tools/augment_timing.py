#!/usr/bin/env python
"""Two-view augmentation: where the time goes (host parameter draws vs the two HIP launches) and how it
compares with the CPU restatement of the reference's per-sample pipeline.
Usage: python tools/augment_timing.py [out.json]"""
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import augment_oracle as A
from peclr_amd import TwoViewAugmenter, _capi

DEV = torch.device("cuda:0")


def main():
    rows = []
    g = np.random.default_rng(0)
    for b, size in ((128, 128), (128, 224), (512, 128)):
        images_np = g.integers(0, 256, (b, 224, 224, 3), dtype=np.uint8)
        images = torch.from_numpy(images_np).to(DEV)
        joints = torch.from_numpy(np.concatenate([g.normal((112, 108), 25, (b, 21, 2)), g.normal(0, 1, (b, 21, 1))], 2)).float()
        aug = TwoViewAugmenter(params={"resize_shape": [size, size]}, rng=random.Random(1))
        for _ in range(3):
            aug(images, joints)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            params, views = aug.sample_batch(joints, (224, 224))
        host_ms = (time.perf_counter() - t0) / 5 * 1e3
        _capi.EVENT_LOG = {}
        t0 = time.perf_counter()
        for _ in range(10):
            aug(images, joints)
        torch.cuda.synchronize()
        total_ms = (time.perf_counter() - t0) / 10 * 1e3
        ev = {k: round(sum(s.elapsed_time(e) for s, e, *_ in v) / len(v) * 1e3, 1) for k, v in _capi.EVENT_LOG.items()}
        _capi.EVENT_LOG = None
        # CPU restatement: one sample, both views (what one DataLoader worker does per item)
        rng = random.Random(1)
        t0 = time.perf_counter()
        n_cpu = 8
        for i in range(n_cpu):
            A.prepare_hybrid2_sample(images_np[i], joints[i].numpy(), aug.flags, aug.params, rng)
        cpu_ms = (time.perf_counter() - t0) / n_cpu * 1e3
        row = {"batch": b, "out": size, "host_param_ms": round(host_ms, 2), "kernel_us": ev,
               "end_to_end_ms": round(total_ms, 2), "images_per_s": round(2 * b / total_ms * 1e3),
               "cpu_restatement_ms_per_sample": round(cpu_ms, 1), "cpu_images_per_s_per_core": round(2 / cpu_ms * 1e3, 1)}
        print(row, flush=True)
        rows.append(row)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
