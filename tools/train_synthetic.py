#!/usr/bin/env python
"""End-to-end functional run: uint8 source images + 2.5D joints -> GPU two-view augmentation -> Hybrid2Model
(ResNet-18, crop+rotate alignment) -> NT-Xent -> LARS(Adam), through Trainer.fit with hipGraph replay.
The "dataset" is a fixed set of structured synthetic images, so instance discrimination is learnable and
the loss must fall.  Usage: python tools/train_synthetic.py [out.json]"""
import json
import os
import random
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import numpy as np
import torch

from peclr_amd import Hybrid2Model, Trainer, TwoViewAugmenter, hybrid2_config
from peclr_amd.bn2d import enable_hip_batchnorm

DEV = torch.device("cuda:0")
N_IMAGES, PAIRS, EPOCHS, SIZE = 512, 64, 12, 128


def make_dataset():
    g = np.random.default_rng(0)
    yy, xx = np.mgrid[0:224, 0:224]
    imgs = []
    for i in range(N_IMAGES):
        f = g.uniform(5, 40, 3)
        ph = g.uniform(0, 6.28, 3)
        base = np.stack([127 + 100 * np.sin(xx / f[0] + ph[0]), 127 + 100 * np.cos(yy / f[1] + ph[1]),
                         127 + 100 * np.sin((xx + yy) / f[2] + ph[2])], axis=2)
        imgs.append(np.clip(base + g.normal(0, 10, base.shape), 0, 255).astype(np.uint8))
    joints = np.concatenate([g.normal((112, 108), 25, (N_IMAGES, 21, 2)), g.normal(0, 1, (N_IMAGES, 21, 1))], axis=2)
    return torch.from_numpy(np.stack(imgs)).to(DEV), torch.from_numpy(joints).float()


def main():
    images, joints = make_dataset()
    cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, augmentation=["crop", "rotate"],
                         batch_size=PAIRS, num_samples=N_IMAGES, warmup_epochs=2, pretrained=False)
    torch.manual_seed(0)
    model = Hybrid2Model(cfg).to(DEV).train()
    model.encoder = model.encoder.to(memory_format=torch.channels_last)
    enable_hip_batchnorm(model.encoder)
    aug = TwoViewAugmenter(params={"resize_shape": [SIZE, SIZE]}, rng=random.Random(0))
    order = random.Random(1)

    def batches(epoch):
        idx = list(range(N_IMAGES))
        order.shuffle(idx)
        for s in range(0, N_IMAGES, PAIRS):
            sel = torch.tensor(idx[s:s + PAIRS])
            yield aug(images[sel.to(DEV)], joints[sel])

    curve = []
    orig = model.training_epoch_end

    def epoch_end(outputs):
        orig(outputs)
        curve.append(round(float(model.train_metrics_epoch["loss"]), 4))
        print(f"epoch {len(curve) - 1}: mean loss {curve[-1]}", flush=True)

    model.training_epoch_end = epoch_end
    tr = Trainer(max_epochs=EPOCHS, hip_graph=True)
    t0 = time.perf_counter()
    tr.fit(model, batches)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"epochs": EPOCHS, "steps": tr.global_step, "pairs_per_step": PAIRS, "seconds": round(dt, 1),
           "images_per_s_incl_augmentation_and_capture": round(2 * PAIRS * tr.global_step / dt), "epoch_mean_loss": curve}
    print(json.dumps(out))
    assert curve[-1] < 0.8 * curve[0], "the loss did not fall"
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
