"""Generate the golden vectors under tests/golden/ from the REFERENCE's own functions.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference is imported under third-party stubs (`_ref_import.py`); its functions are
called on seeded inputs and the inputs + outputs are stored as small .npz / .json
fixtures.  The fixtures are data; no reference source is copied.  The tests never
import the reference.

Fixture index:
  g1_ntxent_N{2,8,32}.npz     vanila_contrastive_loss: z1,z2 -> loss, S, dz1, dz2
  g2_rotate.npz               rotate_encoding / get_rotation_2D_matrix
  g3_translate.npz            translate_encodings
  g4_hybrid2_<aug>.npz        Hybrid2Model.training_step (+backward) with a stand-in encoder
  g5_simclr.npz               SimCLR.contrastive_step (+backward)
  g6_head.npz                 projection head fwd/bwd incl. BN running stats
  g7_optim.json               exclude_from_wt_decay membership, configure_optimizers numbers
  g8_state_dict.json          state_dict key order/shapes of the head
"""
import json
import os
import sys

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import import_reference  # noqa: E402

edict, ref_utils, SimCLR, Hybrid2Model = import_reference()
F = torch.nn.functional


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


# ---------------------------------------------------------------- G1
def g1():
    for n in (2, 8, 32):
        g = torch.Generator().manual_seed(100 + n)
        z1 = F.normalize(torch.randn(n, 128, generator=g)).requires_grad_()
        z2 = F.normalize(torch.randn(n, 128, generator=g)).requires_grad_()
        loss = ref_utils.vanila_contrastive_loss(z1, z2)
        loss.backward()
        z = torch.cat([z1, z2]).detach()
        save(f"g1_ntxent_N{n}.npz", z1=npy(z1), z2=npy(z2), loss=npy(loss),
             sim=npy(z @ z.t()), dz1=npy(z1.grad), dz2=npy(z2.grad),
             temperature=np.float32(0.5))
    # non-default temperature, one case
    g = torch.Generator().manual_seed(7)
    z1 = F.normalize(torch.randn(8, 128, generator=g)).requires_grad_()
    z2 = F.normalize(torch.randn(8, 128, generator=g)).requires_grad_()
    loss = ref_utils.vanila_contrastive_loss(z1, z2, temperature=0.1)
    loss.backward()
    z = torch.cat([z1, z2]).detach()
    save("g1_ntxent_N8_tau01.npz", z1=npy(z1), z2=npy(z2), loss=npy(loss), sim=npy(z @ z.t()),
         dz1=npy(z1.grad), dz2=npy(z2.grad), temperature=np.float32(0.1))


# ---------------------------------------------------------------- G2 / G3
def g2_g3():
    g = torch.Generator().manual_seed(22)
    m = 12
    q = F.normalize(torch.randn(m, 128, generator=g)).view(m, 64, 2)
    angles = torch.tensor([0., 45., -45., 44., 1., -1., 30., -17., 90., 180., 13., -33.],
                          dtype=torch.float64)
    c = q.mean(1)
    rmat = ref_utils.get_rotation_2D_matrix(angles, c[:, 0], c[:, 1], scale=1.0)
    qin = q.clone().requires_grad_()
    out = ref_utils.rotate_encoding(qin * 1.0, angles)  # function mutates its argument
    w = torch.randn(out.shape, generator=g)
    (out * w).sum().backward()
    save("g2_rotate.npz", q=npy(q), angle=npy(angles), out=npy(out), rot_mat=npy(rmat),
         dout=npy(w), dq=npy(qin.grad))

    jx = torch.tensor([0, -14, -7, -1, -3, 5, -14, 0, -2, -9, -11, -6], dtype=torch.int64)
    jy = torch.tensor([-14, 0, -3, -8, 2, -1, -14, 0, -5, -10, -4, -13], dtype=torch.int64)
    for size in (224, 448):
        tx = -(jx / float(size))
        ty = -(jy / float(size))
        out = ref_utils.translate_encodings(q.clone(), tx, ty)
        save(f"g3_translate_{size}.npz", q=npy(q), jitter_x=npy(jx), jitter_y=npy(jy),
             tx=npy(tx), ty=npy(ty), out=npy(out), size=np.int64(size))


# ---------------------------------------------------------------- G4 / G5 / G6
class PoolEncoder(nn.Module):
    """Stand-in for the ResNet: spatial mean of the 'image' -> [M, C] features.

    Keeps the output so the gradient w.r.t. the encoder output can be stored."""

    def forward(self, x):
        self.out = x.mean(dim=(2, 3)).detach().requires_grad_()  # leaf: .grad = dL/dh
        return self.out


def make_model(cls, din, hid, aug, seed):
    cfg = edict({"projection_head_input_dim": din, "projection_head_hidden_dim": hid,
                 "output_dim": 128, "augmentation": aug, "batch_size": 8, "num_samples": 64,
                 "num_of_mini_batch": 1, "lr": 1e-4, "opt_weight_decay": 1e-6,
                 "warmup_epochs": 10, "optimizer": "LARS"})
    torch.manual_seed(seed)
    model = cls(cfg)  # no resnet_size key -> encoder construction skipped (base_model.py:21)
    model.encoder = PoolEncoder()
    # non-trivial BN affine so dgamma/dbeta paths are exercised
    with torch.no_grad():
        model.projection_head[1].weight.uniform_(0.5, 1.5)
        model.projection_head[1].bias.uniform_(-0.2, 0.2)
    model.train()
    return model


def make_batch(n, c, hh, ww, seed, rotate=True):
    g = torch.Generator().manual_seed(seed)
    b = {
        "transformed_image1": torch.randn(n, c, hh, ww, generator=g),
        "transformed_image2": torch.randn(n, c, hh, ww, generator=g),
        "jitter_x_1": torch.randint(-14, 1, (n,), generator=g),
        "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
        "jitter_y_1": torch.randint(-14, 1, (n,), generator=g),
        "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
        "crop_margin_scale_1": torch.rand(n, generator=g, dtype=torch.float64),
        "blur_flag_1": torch.zeros(n, dtype=torch.bool),
    }
    if rotate:
        b["angle_1"] = torch.randint(-45, 46, (n,), generator=g).double()
        b["angle_2"] = torch.randint(-45, 46, (n,), generator=g).double()
    return b


def head_params(model):
    ph = model.projection_head
    return dict(w1=ph[0].weight, b1=ph[0].bias, gamma=ph[1].weight, beta=ph[1].bias,
                w2=ph[3].weight)


def run_step(model, batch, step_fn):
    ph = model.projection_head
    rm0, rv0 = npy(ph[1].running_mean).copy(), npy(ph[1].running_var).copy()
    params = head_params(model)
    w0 = {k: npy(v).copy() for k, v in params.items()}
    out = step_fn(batch)
    loss = out["loss"] if isinstance(out, dict) else out
    loss.backward()
    res = {f"in_{k}": v for k, v in w0.items()}
    res.update({f"d{k}": npy(v.grad) for k, v in params.items()})
    res.update(h=npy(model.encoder.out), dh=npy(model.encoder.out.grad),
               running_mean0=rm0, running_var0=rv0,
               running_mean1=npy(ph[1].running_mean), running_var1=npy(ph[1].running_var),
               num_batches_tracked1=npy(ph[1].num_batches_tracked), loss=npy(loss))
    if isinstance(out, dict):
        res.update({f"out_{k}": npy(v) for k, v in out.items()})
        res["out_keys"] = np.array(list(out.keys()))
    return res


def g4_g5_g6():
    cases = {"none": [], "crop": ["crop"], "rotate": ["rotate"], "crop_rotate": ["crop", "rotate"]}
    for i, (tag, aug) in enumerate(cases.items()):
        n, c, hh, ww = 6, 48, 4, 8  # H != W exercises the x/shape[0], y/shape[1] quirk
        model = make_model(Hybrid2Model, c, 96, aug, seed=40 + i)
        batch = make_batch(n, c, hh, ww, seed=50 + i)
        res = run_step(model, batch, lambda b: model.training_step(b, 0))
        # z1,z2 as the reference forms them (fresh forward in eval-free mode is not possible
        # without touching BN stats again, so recompute from a cloned model)
        res.update({f"batch_{k}": npy(v) for k, v in batch.items() if "image" not in k})
        res.update(image_hw=np.array([hh, ww], np.int64), n_pairs=np.int64(n),
                   plot_param_keys=np.array(list(model.plot_params["params"].keys())))
        save(f"g4_hybrid2_{tag}.npz", **res)

    # realistic head width, square 224-like extent emulated by (7,7) 'images'
    model = make_model(Hybrid2Model, 64, 512, ["crop", "rotate"], seed=61)
    batch = make_batch(16, 64, 7, 7, seed=62)
    res = run_step(model, batch, lambda b: model.training_step(b, 0))
    res.update({f"batch_{k}": npy(v) for k, v in batch.items() if "image" not in k})
    res.update(image_hw=np.array([7, 7], np.int64), n_pairs=np.int64(16))
    save("g4_hybrid2_wide.npz", **res)

    # validation_step: {"loss"} only, but train_metrics still gets the 16 stats (quirk)
    model = make_model(Hybrid2Model, 48, 96, ["crop", "rotate"], seed=63)
    batch = make_batch(4, 48, 4, 4, seed=64)
    out = model.validation_step(batch, 0)
    save("g4_hybrid2_val.npz", h=npy(model.encoder.out), loss=npy(out["loss"]),
         out_keys=np.array(list(out.keys())),
         train_metric_keys=np.array(list(model.train_metrics.keys())),
         **{f"in_{k}": npy(v) for k, v in head_params(model).items()},
         **{f"batch_{k}": npy(v) for k, v in batch.items() if "image" not in k},
         image_hw=np.array([4, 4], np.int64), n_pairs=np.int64(4))

    # G5 SimCLR
    model = make_model(SimCLR, 48, 96, [], seed=70)
    batch = make_batch(6, 48, 4, 4, seed=71, rotate=False)
    res = run_step(model, batch, lambda b: model.contrastive_step(b))
    res.update(n_pairs=np.int64(6))
    save("g5_simclr.npz", **res)

    # G6 head alone, two consecutive train-mode forwards (running stats twice)
    model = make_model(SimCLR, 40, 72, [], seed=80)
    ph = model.projection_head
    g = torch.Generator().manual_seed(81)
    h = torch.randn(10, 40, generator=g).requires_grad_()
    w0 = {f"in_{k}": npy(v).copy() for k, v in head_params(model).items()}
    p = ph(h)
    dp = torch.randn(p.shape, generator=g)
    (p * dp).sum().backward()
    rm1, rv1 = npy(ph[1].running_mean).copy(), npy(ph[1].running_var).copy()
    h2 = torch.randn(10, 40, generator=g)
    p2 = ph(h2)
    save("g6_head.npz", h=npy(h), p=npy(p), dp=npy(dp), dh=npy(h.grad), h2=npy(h2), p2=npy(p2),
         running_mean1=rm1, running_var1=rv1, running_mean2=npy(ph[1].running_mean),
         running_var2=npy(ph[1].running_var), num_batches_tracked2=npy(ph[1].num_batches_tracked),
         **w0, **{f"d{k}": npy(v.grad) for k, v in head_params(model).items()})


# ---------------------------------------------------------------- G7 / G8
def g7_g8():
    import sys as _sys

    rec = {}

    class LARSRec:
        def __init__(self, optimizer):
            self.optim = optimizer
            self.param_groups = optimizer.param_groups
            rec["lars_wrapped"] = type(optimizer).__name__

    class SchedRec:
        def __init__(self, optimizer, warmup_epochs, max_epochs, warmup_start_lr, eta_min):
            rec.update(warmup_epochs=warmup_epochs, max_epochs=max_epochs,
                       warmup_start_lr=warmup_start_lr, eta_min=eta_min)

    import src.models.base_model as bm

    bm.LARSWrapper = LARSRec
    bm.LinearWarmupCosineAnnealingLR = SchedRec

    names = [
        "encoder.features.0.weight", "encoder.features.1.weight", "encoder.features.1.bias",
        "encoder.features.4.0.conv1.weight", "encoder.features.4.0.bn1.weight",
        "encoder.features.4.0.bn1.bias", "encoder.features.4.0.downsample.0.weight",
        "encoder.features.4.0.downsample.1.weight", "encoder.features.4.0.downsample.1.bias",
        "encoder.final_layer.0.weight", "encoder.final_layer.0.bias",
        "projection_head.0.weight", "projection_head.0.bias", "projection_head.1.weight",
        "projection_head.1.bias", "projection_head.3.weight",
    ]
    model = make_model(Hybrid2Model, 48, 96, [], seed=90)
    named = [(n, nn.Parameter(torch.zeros(1))) for n in names]
    groups = model.exclude_from_wt_decay(iter(named), weight_decay=1e-6)
    ids = {id(p): n for n, p in named}
    membership = {"decay": [ids[id(p)] for p in groups[0]["params"]],
                  "no_decay": [ids[id(p)] for p in groups[1]["params"]],
                  "weight_decay": [groups[0]["weight_decay"], groups[1]["weight_decay"]]}

    optim_cases = []
    for accum, batch_size, num_samples, world, max_ep, lr_max in (
            (1, 128, 32560 + 44994, 1, 100, None), (16, 128, 32560, 1, 100, None),
            (1, 128, 100000, 8, 50, None), (4, 64, 5000, 1, 100, 40)):
        rec.clear()
        model = make_model(Hybrid2Model, 48, 96, [], seed=91)
        model.config.batch_size = batch_size
        model.config.num_samples = num_samples
        model.config.num_of_mini_batch = accum
        if lr_max is not None:
            model.config["lr_max_epochs"] = lr_max
        model.trainer = type("T", (), {"world_size": world, "max_epochs": max_ep})()
        model.setup("fit")
        opts, scheds = model.configure_optimizers()
        adam = opts[0].optim
        optim_cases.append(dict(
            accum=accum, batch_size=batch_size, num_samples=num_samples, world_size=world,
            trainer_max_epochs=max_ep, lr_max_epochs=lr_max,
            train_iters_per_epoch=model.train_iters_per_epoch,
            lr=[g["lr"] for g in adam.param_groups],
            weight_decay=[g["weight_decay"] for g in adam.param_groups],
            betas=list(adam.param_groups[0]["betas"]), eps=adam.param_groups[0]["eps"],
            n_params=[len(g["params"]) for g in adam.param_groups],
            sched_keys={k: v for k, v in scheds[0].items() if k != "scheduler"}, **rec))
    # non-LARS branch: CosineAnnealingLR(T_max=max_epochs)
    model = make_model(Hybrid2Model, 48, 96, [], seed=92)
    model.config.optimizer = "adam"
    model.trainer = type("T", (), {"world_size": 1, "max_epochs": 10})()
    model.setup("fit")
    opts, scheds = model.configure_optimizers()
    cosine = dict(opt_type=type(opts[0]).__name__, sched_type=type(scheds[0]["scheduler"]).__name__,
                  T_max=scheds[0]["scheduler"].T_max)
    with open(os.path.join(HERE, "g7_optim.json"), "w") as f:
        json.dump(dict(membership=membership, cases=optim_cases, cosine=cosine), f, indent=1)
    print("wrote g7_optim.json")

    model = make_model(Hybrid2Model, 2048, 512, [], seed=93)
    sd = {k: list(v.shape) for k, v in model.state_dict().items()}
    epoch_end = {}
    outs = [{"loss": torch.tensor(1.0), "a": torch.tensor(2.0)},
            {"loss": torch.tensor(3.0), "a": torch.tensor(6.0)}]
    model.training_epoch_end(outs)
    epoch_end["train_metrics_epoch"] = {k: float(v) for k, v in model.train_metrics_epoch.items()}
    epoch_end["logged"] = {k: float(v) for k, v in model.logged.items()}
    model.validation_epoch_end(outs)
    epoch_end["validation_metrics_epoch"] = {k: float(v)
                                             for k, v in model.validation_metrics_epoch.items()}
    with open(os.path.join(HERE, "g8_state_dict.json"), "w") as f:
        json.dump(dict(head_state_dict=sd, epoch_end=epoch_end,
                       init_attrs=sorted(k for k in ("train_metrics_epoch", "train_metrics",
                                                     "validation_metrics_epoch", "plot_params")
                                         if hasattr(model, k))), f, indent=1)
    print("wrote g8_state_dict.json")


if __name__ == "__main__":
    torch.set_num_threads(1)  # deterministic summation order in the captured vectors
    g1()
    g2_g3()
    g4_g5_g6()
    g7_g8()
