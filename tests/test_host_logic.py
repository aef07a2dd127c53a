"""Host logic above the C ABI, on CPU: module surface, autograd wiring, optimiser plumbing,
trainer, checkpoint export.  The kernels are replaced by the oracle-backed test double
(tests/_capi_double.py); goldens come from the reference."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import peclr_oracle as O
from tests import _capi_double
from tests.conftest import GOLDEN, load_golden


@pytest.fixture(autouse=True)
def cpu_kernels(monkeypatch):
    _capi_double.install(monkeypatch)


class FixedEncoder(torch.nn.Module):
    def __init__(self, h):
        super().__init__()
        self.h = torch.nn.Parameter(torch.from_numpy(h))

    def forward(self, x):
        return self.h


def build_model(cls, g, aug):
    from peclr_amd import Config

    cfg = Config(projection_head_input_dim=g["in_w1"].shape[1], projection_head_hidden_dim=g["in_w1"].shape[0],
                 output_dim=128, augmentation=aug, batch_size=8, num_samples=64, num_of_mini_batch=1, lr=1e-4,
                 opt_weight_decay=1e-6, warmup_epochs=10, optimizer="LARS")
    model = cls(cfg)
    model.encoder = FixedEncoder(g["h"])
    ph = model.projection_head
    with torch.no_grad():
        for t, k in ((ph[0].weight, "in_w1"), (ph[0].bias, "in_b1"), (ph[1].weight, "in_gamma"),
                     (ph[1].bias, "in_beta"), (ph[3].weight, "in_w2")):
            t.copy_(torch.from_numpy(g[k]))
    return model.train()


def golden_batch(g):
    n = int(g["n_pairs"])
    hh, ww = (int(v) for v in g["image_hw"]) if "image_hw" in g else (4, 4)
    b = {"transformed_image1": torch.zeros(n, 1, hh, ww), "transformed_image2": torch.zeros(n, 1, hh, ww)}
    for k in g:
        if k.startswith("batch_"):
            b[k[6:]] = torch.from_numpy(g[k])
    return b


@pytest.mark.parametrize("tag,aug", [("none", []), ("crop_rotate", ["crop", "rotate"])])
def test_training_step_surface_and_autograd_wiring(tag, aug):
    from peclr_amd import Hybrid2Model

    g = load_golden(f"g4_hybrid2_{tag}.npz")
    model = build_model(Hybrid2Model, g, aug)
    out = model.training_step(golden_batch(g), 0)
    assert list(out.keys()) == [str(k) for k in g["out_keys"]]
    assert abs(float(out["loss"]) - float(g["loss"])) < 5e-6
    for k in out:
        assert out[k].dim() == 0
        assert abs(float(out[k]) - float(g[f"out_{k}"])) < 5e-6
    assert out["loss"].requires_grad and not out["proj1x_mean"].requires_grad
    out["loss"].backward()
    ph = model.projection_head
    for k, v in dict(dh=model.encoder.h.grad, dw1=ph[0].weight.grad, dgamma=ph[1].weight.grad,
                     dbeta=ph[1].bias.grad, dw2=ph[3].weight.grad).items():
        np.testing.assert_allclose(v.numpy(), g[k], rtol=0, atol=3e-5 * max(1.0, float(np.abs(g[k]).max())))
    np.testing.assert_allclose(ph[1].running_var.numpy(), g["running_var1"], atol=1e-6)
    assert set(model.plot_params) == {"image1", "image2", "params"}


def test_get_transformed_projections_and_forward():
    from peclr_amd import Hybrid2Model

    g = load_golden("g4_hybrid2_crop_rotate.npz")
    model = build_model(Hybrid2Model, g, ["crop", "rotate"])
    z1, z2 = model.get_transformed_projections(golden_batch(g))
    n = int(g["n_pairs"])
    assert z1.shape == (n, 128) and z2.shape == (n, 128)
    np.testing.assert_allclose(z1.detach().norm(dim=1).numpy(), 1.0, atol=1e-6)
    assert list(model.train_metrics.keys()) == list(O.stat_keys())
    model.eval()
    out = model(torch.zeros(2 * n, 1, 4, 4))
    assert set(out) == {"embedding", "projection"} and out["projection"].shape == (2 * n, 128)


def test_state_dict_contract():
    import warnings

    from peclr_amd import Hybrid2Model, hybrid2_config

    with open(os.path.join(GOLDEN, "g8_state_dict.json")) as f:
        g = json.load(f)
    warnings.simplefilter("ignore")
    model = Hybrid2Model(hybrid2_config(resnet_size="18", projection_head_input_dim=2048))
    sd = model.state_dict()
    head = {k: list(v.shape) for k, v in sd.items() if k.startswith("projection_head")}
    assert list(head.items()) == list(g["head_state_dict"].items())  # names, order and shapes
    keys = list(sd)
    assert keys[0] == "encoder.features.0.weight" and keys[1] == "encoder.features.1.weight"
    assert keys.index("encoder.final_layer.0.bias") + 1 == keys.index("projection_head.0.weight")
    for attr in g["init_attrs"]:
        assert hasattr(model, attr)
    outs = [{"loss": torch.tensor(1.0), "a": torch.tensor(2.0)}, {"loss": torch.tensor(3.0), "a": torch.tensor(6.0)}]
    model.training_epoch_end(outs)
    assert {k: float(v) for k, v in model.train_metrics_epoch.items()} == g["epoch_end"]["train_metrics_epoch"]
    assert {k: float(v) for k, v in model.logged.items()} == g["epoch_end"]["logged"]
    model.validation_epoch_end(outs)
    assert {k: float(v) for k, v in model.validation_metrics_epoch.items()} == \
        g["epoch_end"]["validation_metrics_epoch"]


def test_configure_optimizers_matches_reference_numbers():
    import warnings

    from peclr_amd import Hybrid2Model, hybrid2_config
    from peclr_amd.optim import LARSAdam, LinearWarmupCosineAnnealingLR

    with open(os.path.join(GOLDEN, "g7_optim.json")) as f:
        g = json.load(f)
    warnings.simplefilter("ignore")
    for c in g["cases"]:
        cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, batch_size=c["batch_size"],
                             num_samples=c["num_samples"], num_of_mini_batch=c["accum"])
        if c["lr_max_epochs"] is not None:
            cfg["lr_max_epochs"] = c["lr_max_epochs"]
        model = Hybrid2Model(cfg)
        model.trainer = type("T", (), {"world_size": c["world_size"], "max_epochs": c["trainer_max_epochs"]})()
        model.setup("fit")
        assert model.train_iters_per_epoch == c["train_iters_per_epoch"]
        (opt,), (sched,) = model.configure_optimizers()
        assert isinstance(opt, LARSAdam) and opt.lars
        assert [grp["lr"] for grp in opt.param_groups] == pytest.approx(c["lr"], rel=1e-12) or \
            [grp["initial_lr"] for grp in opt.param_groups] == pytest.approx(c["lr"], rel=1e-12)
        assert [grp["weight_decay"] for grp in opt.param_groups] == c["weight_decay"]
        assert list(opt.param_groups[0]["betas"]) == c["betas"] and opt.param_groups[0]["eps"] == c["eps"]
        s = sched["scheduler"]
        assert isinstance(s, LinearWarmupCosineAnnealingLR)
        assert (s.warmup_epochs, s.max_epochs, s.warmup_start_lr, s.eta_min) == \
            (c["warmup_epochs"], c["max_epochs"], c["warmup_start_lr"], c["eta_min"])
        assert {k: v for k, v in sched.items() if k != "scheduler"} == c["sched_keys"]
    # weight-decay membership incl. the substring quirks
    names = g["membership"]["decay"] + g["membership"]["no_decay"]
    named = [(n, torch.nn.Parameter(torch.zeros(1))) for n in names]
    groups = model.exclude_from_wt_decay(iter(named), weight_decay=1e-6)
    ids = {id(p): n for n, p in named}
    assert [ids[id(p)] for p in groups[0]["params"]] == g["membership"]["decay"]
    assert [ids[id(p)] for p in groups[1]["params"]] == g["membership"]["no_decay"]
    # non-LARS branch
    model.config.optimizer = "adam"
    (opt,), (sched,) = model.configure_optimizers()
    assert not opt.lars and type(sched["scheduler"]).__name__ == g["cosine"]["sched_type"]


def test_lars_adam_foreach_matches_oracle_and_schedule():
    from peclr_amd.optim import LARSAdam, LinearWarmupCosineAnnealingLR

    rng = np.random.default_rng(3)
    p0 = rng.standard_normal((40, 30)).astype(np.float32)
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    zero = torch.nn.Parameter(torch.zeros(7))
    opt = LARSAdam([{"params": [p, zero], "weight_decay": 1e-6}], lr=2e-3, lars=True, fused=False)
    sched = LinearWarmupCosineAnnealingLR(opt, warmup_epochs=4, max_epochs=10)
    pr, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(8):
        lr = O.warmup_cosine_lr(step, 2e-3, 4, 10)
        assert opt.param_groups[0]["lr"] == pytest.approx(lr, abs=1e-12)
        g = rng.standard_normal(p0.shape).astype(np.float32)
        p.grad = torch.from_numpy(g.copy())
        zero.grad = torch.from_numpy(rng.standard_normal(7).astype(np.float32))
        z_before = zero.detach().clone()
        opt.step()
        sched.step()
        pr, m, v, _ = O.lars_adam_step(pr, g, m, v, step + 1, lr, 1e-6)
        np.testing.assert_allclose(p.detach().numpy(), pr, atol=1e-6)
        if lr == 0:
            assert torch.equal(zero.detach(), z_before)
    with pytest.raises(Exception):
        LARSAdam([p], fused=True)  # fused needs HIP tensors: no silent fallback


def test_trainer_accumulation_and_checkpoint_roundtrip(tmp_path, monkeypatch):
    import warnings

    from peclr_amd import Hybrid2Model, Trainer, hybrid2_config, peclr_to_torchvision, get_encoder_state_dict
    from peclr_amd import resnet

    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    n = 2
    cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, augmentation=["crop", "rotate"],
                         batch_size=n, num_samples=8, num_of_mini_batch=2, pretrained=False)
    model = Hybrid2Model(cfg)

    def batches(epoch):
        g = torch.Generator().manual_seed(epoch)
        for _ in range(4):
            yield {"transformed_image1": torch.randn(n, 3, 32, 32, generator=g),
                   "transformed_image2": torch.randn(n, 3, 32, 32, generator=g),
                   "jitter_x_1": torch.randint(-14, 1, (n,), generator=g),
                   "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
                   "jitter_y_1": torch.randint(-14, 1, (n,), generator=g),
                   "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
                   "angle_1": torch.randint(-45, 46, (n,), generator=g).double(),
                   "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}

    # the reference's layout: $SAVED_MODELS_BASE_PATH/<experiment>/checkpoints/epoch=K.ckpt (utils.py:189-206)
    monkeypatch.setenv("SAVED_MODELS_BASE_PATH", str(tmp_path))
    ckpt_dir = tmp_path / "exp1" / "checkpoints"
    tr = Trainer(max_epochs=2, accumulate_grad_batches=2, checkpoint_dir=str(ckpt_dir), save_top_k=1)
    tr.fit(model, batches, val_batches=lambda e: list(batches(e))[:1])
    assert tr.global_step == 4  # 2 epochs x 4 micro-batches / accumulate 2
    assert set(model.train_metrics_epoch) == set(O.stat_keys()) | {"loss"}
    assert "checkpoint_saving_loss" in model.logged and list(model.validation_metrics_epoch) == ["loss"]
    ckpts = os.listdir(ckpt_dir)
    assert len(ckpts) == 1 and ckpts[0].startswith("epoch=")
    path = os.path.join(ckpt_dir, ckpts[0])
    # export: every `features` entry lands in a torchvision-layout ResNet, fc untouched
    target = resnet.resnet18()
    fc_before = target.fc.weight.detach().clone()
    peclr_to_torchvision(target, path)
    saved = torch.load(path, map_location="cpu")["state_dict"]  # top-k keeps the best epoch, not the last
    tsd = target.state_dict()
    assert torch.equal(tsd["conv1.weight"], saved["encoder.features.0.weight"])
    assert torch.equal(tsd["layer4.1.bn2.running_var"], saved["encoder.features.7.1.bn2.running_var"])
    assert list(saved) == list(model.state_dict())
    assert torch.equal(target.fc.weight, fc_before)
    from peclr_amd import get_latest_checkpoint

    assert get_latest_checkpoint("exp1") == path and get_latest_checkpoint("exp1", "epoch=7.ckpt").endswith("epoch=7.ckpt")
    enc = get_encoder_state_dict("exp1", "")   # (saved_model_path, checkpoint), as utils.py:209-225
    assert list(enc)[0] == "features.0.weight" and "final_layer.0.bias" in enc
    assert all(not k.startswith("encoder.") and "projection_head" not in k for k in enc)
    with pytest.raises(Exception, match="not of type ResNet"):
        peclr_to_torchvision(torch.nn.Linear(2, 2), path)


def test_resume_continues_the_run_exactly(tmp_path, monkeypatch):
    """Trainer.resume: weights, optimiser moments/step counts, schedule position, epoch and global step;
    a run resumed after epoch 0 ends where the uninterrupted run ends.  restore_model: weights only
    (experiments/utils.py:535-546)."""
    import copy
    import warnings

    from peclr_amd import Hybrid2Model, Trainer, hybrid2_config, restore_model

    warnings.simplefilter("ignore")
    torch.manual_seed(3)
    n = 2
    cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, augmentation=["crop", "rotate"],
                         batch_size=n, num_samples=6, warmup_epochs=1, pretrained=False)
    base = Hybrid2Model(cfg)

    def batches(epoch):
        g = torch.Generator().manual_seed(50 + epoch)
        for _ in range(3):
            yield {"transformed_image1": torch.randn(n, 3, 32, 32, generator=g),
                   "transformed_image2": torch.randn(n, 3, 32, 32, generator=g),
                   "jitter_x_1": torch.randint(-14, 1, (n,), generator=g), "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
                   "jitter_y_1": torch.randint(-14, 1, (n,), generator=g), "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
                   "angle_1": torch.randint(-45, 46, (n,), generator=g).double(),
                   "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}

    straight = copy.deepcopy(base)
    monkeypatch.setenv("SAVED_MODELS_BASE_PATH", str(tmp_path))
    ts = Trainer(max_epochs=2, checkpoint_dir=str(tmp_path / "a" / "checkpoints"), save_top_k=5)
    ts.fit(straight, batches)
    # the "interrupted" run is the same run cut after epoch 0: its checkpoint is a/epoch=0.ckpt
    resumed = copy.deepcopy(base)
    tr = Trainer(max_epochs=2, checkpoint_dir=str(tmp_path / "b" / "checkpoints"), save_top_k=5).attach(resumed)
    tr.resume(os.path.join(tmp_path / "a" / "checkpoints", "epoch=0.ckpt"))
    assert tr.global_step == 3 and tr.current_epoch == 1
    tr.fit(resumed, batches)
    assert tr.global_step == ts.global_step == 6
    assert tr.scheduler.last_epoch == ts.scheduler.last_epoch
    for (k, a), (_, b) in zip(straight.state_dict().items(), resumed.state_dict().items()):
        assert torch.equal(a, b), k
    fresh = restore_model(copy.deepcopy(base), "b")                          # (model, experiment_key): newest = epoch=1
    assert torch.equal(fresh.projection_head[3].weight, resumed.projection_head[3].weight)
    again = restore_model(copy.deepcopy(base), "a", "epoch=0.ckpt")
    assert not torch.equal(again.projection_head[3].weight, resumed.projection_head[3].weight)
    assert not torch.equal(again.projection_head[3].weight, base.projection_head[3].weight)


def test_resnet_state_dict_layout():
    from peclr_amd import resnet

    for name, n_entries, n_params in (("resnet18", 122, 11689512), ("resnet50", 320, 25557032),
                                      ("resnet152", 932, 60192808)):
        m = getattr(resnet, name)()
        sd = m.state_dict()
        assert len(sd) == n_entries, name                      # torchvision's state_dict length
        assert sum(p.numel() for p in m.parameters()) == n_params, name   # torchvision's parameter count
        keys = list(sd)
        assert keys[:6] == ["conv1.weight", "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var",
                            "bn1.num_batches_tracked"]
        assert keys[-2:] == ["fc.weight", "fc.bias"]
    b = resnet.resnet50().layer1[0]
    assert [n for n, _ in b.named_children()] == ["conv1", "bn1", "conv2", "bn2", "conv3", "bn3", "relu",
                                                  "downsample"]
    assert b.conv2.stride == (1, 1) and resnet.resnet50().layer2[0].conv2.stride == (2, 2)  # v1.5
    x = torch.randn(2, 3, 64, 64)
    assert resnet.resnet18()(x).shape == (2, 1000)
    assert abs(resnet.conv_flops_per_image(resnet.resnet50()) / 1e9 - 8.18) < 0.05


def _tiny_batches(n, count, seed=0, size=32):
    def batches(epoch):
        g = torch.Generator().manual_seed(seed + epoch)
        for _ in range(count):
            yield {"transformed_image1": torch.randn(n, 3, size, size, generator=g),
                   "transformed_image2": torch.randn(n, 3, size, size, generator=g),
                   "jitter_x_1": torch.randint(-14, 1, (n,), generator=g), "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
                   "jitter_y_1": torch.randint(-14, 1, (n,), generator=g), "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
                   "angle_1": torch.randint(-45, 46, (n,), generator=g).double(),
                   "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}
    return batches


def test_training_epoch_end_monitors_loss_3d_when_the_step_reports_one():
    """base_model.py:111-115: `checkpoint_saving_loss` is the epoch mean of "loss_3d" if the step dicts carry that key
    (the supervised subclasses of the reference inherit the hook), of "loss" otherwise."""
    from peclr_amd.module import BaseModel

    model = BaseModel.__new__(BaseModel)
    torch.nn.Module.__init__(model)
    logged = {}
    model.log = lambda key, value, **kw: logged.__setitem__(key, float(value))
    outs = [{"loss": torch.tensor(1.0), "loss_3d": torch.tensor(5.0)}, {"loss": torch.tensor(3.0), "loss_3d": torch.tensor(9.0)}]
    model.training_epoch_end(outs)
    assert logged["checkpoint_saving_loss"] == 7.0 and float(model.train_metrics_epoch["loss"]) == 2.0
    model.training_epoch_end([{"loss": torch.tensor(1.0)}, {"loss": torch.tensor(3.0)}])
    assert logged["checkpoint_saving_loss"] == 2.0


def test_accumulation_steps_on_the_final_batch_of_the_epoch():
    """Lightning 1.0.8: `should_accumulate = not (accumulation_done or is_final_batch)` -- 5 batches with
    accumulate_grad_batches=2 are 3 optimiser steps per epoch (2 + 2 + 1), the trailing partial window is
    applied (still divided by k) and nothing leaks into the next epoch."""
    import copy
    import warnings

    from peclr_amd import Hybrid2Model, Trainer, hybrid2_config

    warnings.simplefilter("ignore")
    torch.manual_seed(1)
    cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, augmentation=["crop"], batch_size=2,
                         num_samples=20, num_of_mini_batch=2, pretrained=False)
    model = Hybrid2Model(cfg)
    before_last = {}
    tr = Trainer(max_epochs=1, accumulate_grad_batches=2)
    batches = _tiny_batches(2, 5)
    tr.fit(model, batches)
    assert tr.global_step == 3 and tr.scheduler.last_epoch == 3
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0.0 for p in model.parameters())
    # the last step is exactly "one micro-batch / k": replay it by hand from the state after two full windows
    torch.manual_seed(1)
    ref = Hybrid2Model(cfg)
    tr2 = Trainer(max_epochs=1, accumulate_grad_batches=2).attach(ref)
    tr2.zero_grad()
    bl = list(batches(0))
    for i in range(4):
        tr2.training_micro_step(bl[i], i)
    assert tr2.global_step == 2
    out = ref.training_step(bl[4], 4)
    (out["loss"] / 2).backward()
    tr2.optimizer.step()
    for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k


def test_precision_16_uses_a_grad_scaler_and_skips_on_overflow():
    """precision=16 (the reference's default, training_config.json:9) = fp16 autocast + dynamic loss
    scaling: an overflowing step is skipped (weights untouched, scale halved, LR schedule still advances),
    a normal one is applied with unscaled gradients."""
    import warnings

    from peclr_amd import Trainer

    warnings.simplefilter("ignore")

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(4))
            self.trainer, self.blow = None, False

        def setup(self, stage):
            pass

        def configure_optimizers(self):
            from peclr_amd.optim import LARSAdam

            opt = LARSAdam([{"params": [self.w], "weight_decay": 0.0}], lr=1e-2, lars=False, fused=False)
            return [opt], [{"scheduler": torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)}]

        def training_step(self, batch, idx):
            loss = (self.w * batch["transformed_image1"]).sum()
            return {"loss": loss * float("inf") if self.blow else loss}

    m = Tiny()
    tr = Trainer(precision=16).attach(m)
    assert tr.precision == "fp16"
    tr.zero_grad()
    b = {"transformed_image1": torch.arange(4.0)}
    tr.training_micro_step(b, 0)
    scale0 = tr._scaler.get_scale()
    w1 = m.w.detach().clone()
    assert not torch.equal(w1, torch.ones(4)) and float(w1[0]) == 1.0      # the step was taken; a zero gradient moves nothing
    assert torch.allclose(w1[1:], torch.ones(3) - 1e-2, atol=1e-6)   # Adam's first step = -lr*sign(g): unscaled g
    m.blow = True
    tr.training_micro_step(b, 1)
    assert torch.equal(m.w.detach(), w1) and tr._scaler.get_scale() == scale0 / 2
    assert tr.global_step == 2 and tr.scheduler.last_epoch == 2
    with pytest.raises(ValueError):
        Trainer(precision="fp8")


def test_lars_write_back_leaves_the_scaled_gradient_like_the_reference_wrapper():
    from peclr_amd.optim import LARSAdam, LARSWrapper

    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(5, 3))
    g = torch.randn(5, 3)
    p.grad = g.clone()
    opt = LARSWrapper(torch.optim.Adam([{"params": [p], "weight_decay": 1e-2}], lr=0.1))
    assert isinstance(opt, LARSAdam) and opt.write_back
    w0 = p.detach().clone()
    opt.step()
    pn, gn = float(w0.norm()), float(g.norm())
    trust = min(0.02 * pn / (gn + 1e-2 * pn + 1e-8) / 0.1, 1.0)
    np.testing.assert_allclose(p.grad.numpy(), ((g + 1e-2 * w0) * trust).numpy(), rtol=1e-6, atol=1e-8)
    q = torch.nn.Parameter(w0.clone())
    q.grad = g.clone()
    LARSAdam([{"params": [q], "weight_decay": 1e-2}], lr=0.1, fused=False).step()   # default: grad untouched
    assert torch.equal(q.grad, g) and torch.equal(q.detach(), p.detach())


def test_forward_strict_reference_runs_the_encoder_twice():
    """simclr_model.py:54-57 evaluates the encoder a second time for "embedding"; opt-in here."""
    from peclr_amd import Config, SimCLR

    calls = []

    class Enc(torch.nn.Module):
        def forward(self, x):
            calls.append(1)
            return x.flatten(1)

    for strict, expect in ((False, 1), (True, 2)):
        cfg = Config(projection_head_input_dim=12, projection_head_hidden_dim=8, output_dim=128, strict_reference=strict)
        m = SimCLR(cfg)
        m.encoder = Enc()
        m.eval()
        calls.clear()
        out = m(torch.randn(3, 3, 2, 2))
        assert len(calls) == expect and set(out) == {"embedding", "projection"}
        assert out["embedding"].shape == (3, 12) and out["projection"].shape == (3, 128)


def test_peclr_to_torchvision_reports_and_stops_on_a_mismatch(tmp_path, capsys):
    """port_model.py:35-45: key-suffix mismatch or incompatible shapes -> message, no exception, the rest
    of the ResNet untouched."""
    from peclr_amd import peclr_to_torchvision, resnet

    src = resnet.resnet18()
    sd = {"encoder.features." + k: v for k, v in src.state_dict().items() if not k.startswith("fc.")}
    path = str(tmp_path / "w.ckpt")
    torch.save({"state_dict": sd}, path)
    dst = resnet.resnet34()        # same stem, deeper layers: shapes diverge later on
    stem_before = dst.conv1.weight.detach().clone()
    l4_before = dst.layer4[2].conv1.weight.detach().clone()
    peclr_to_torchvision(dst, path)
    assert torch.equal(dst.conv1.weight, src.conv1.weight) and not torch.equal(dst.conv1.weight, stem_before)
    assert torch.equal(dst.layer4[2].conv1.weight, l4_before)
    printed = capsys.readouterr().out
    assert "don't match" in printed or "not compatible" in printed
    # a key list whose suffixes disagree at position 1
    bad = dict([list(sd.items())[0], list(sd.items())[2]])   # conv1.weight, bn1.bias  vs  conv1.weight, bn1.weight
    torch.save({"state_dict": bad}, path)
    peclr_to_torchvision(resnet.resnet18(), path)
    assert "PeCLR layers don't match" in capsys.readouterr().out


def test_activation_checkpointing_is_exact_and_moves_running_stats_once():
    """residual blocks keep only their input and re-run in the backward pass.  Same
    outputs, same gradients, and the BatchNorm running statistics / num_batches_tracked move ONCE per step."""
    import copy
    import warnings

    from peclr_amd import Hybrid2Model, Trainer, hybrid2_config, resnet

    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    plain = resnet.resnet50().train()
    ckpt = copy.deepcopy(plain)
    assert resnet.set_activation_checkpointing(ckpt) == 16
    x = torch.randn(2, 3, 64, 64)
    ya, yb = plain(x), ckpt(x)
    assert torch.equal(ya, yb)
    ya.square().mean().backward()
    yb.square().mean().backward()
    for (n, p), (_, q) in zip(plain.named_parameters(), ckpt.named_parameters()):
        assert torch.equal(p.grad, q.grad), n
    for (n, p), (_, q) in zip(plain.named_buffers(), ckpt.named_buffers()):
        assert torch.equal(p, q), n
    assert int(ckpt.layer3[2].bn2.num_batches_tracked) == 1
    with torch.no_grad():                      # no gradient -> no checkpoint machinery, same numbers
        assert torch.equal(ckpt(x), plain(x))
    # through the trainer flag
    cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, augmentation=["crop"], batch_size=2,
                         num_samples=8, pretrained=False)
    torch.manual_seed(1)
    a = Hybrid2Model(cfg)
    b = copy.deepcopy(a)
    batches = _tiny_batches(2, 2, seed=9)
    Trainer(max_epochs=1).fit(a, batches)
    tr = Trainer(max_epochs=1, activation_checkpointing=True)
    tr.fit(b, batches)
    assert all(m.checkpoint for m in b.encoder.modules() if isinstance(m, resnet.BasicBlock))
    for (k, u), (_, v) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(u, v), k


def test_static_batch_keeps_the_stacked_views_relation():
    """Graph replay copies every new batch into static buffers.  When both views arrive as halves of ONE
    [2N,...] tensor (`transformed_images`: no per-step concatenation), the static copy must keep that relation, and
    a later batch may arrive stacked or as two separate tensors."""
    from peclr_amd import Trainer
    from peclr_amd.module import SimCLR

    n = 3
    stacked = torch.arange(2 * n * 4, dtype=torch.float32).view(2 * n, 1, 2, 2)
    batch = {"transformed_images": stacked, "transformed_image1": stacked[:n], "transformed_image2": stacked[n:],
             "angle_1": torch.zeros(n, dtype=torch.float64)}
    static = Trainer._clone_batch(batch)
    assert static["transformed_images"].data_ptr() != stacked.data_ptr()
    assert static["transformed_image1"].data_ptr() == static["transformed_images"].data_ptr()
    assert static["transformed_image2"].data_ptr() == static["transformed_images"][n:].data_ptr()
    assert SimCLR._two_views(static) is static["transformed_images"]              # no torch.cat
    # stacked -> stacked
    new = {k: v + 100 if v.is_floating_point() else v for k, v in batch.items()}
    new["transformed_image1"], new["transformed_image2"] = new["transformed_images"][:n], new["transformed_images"][n:]
    Trainer._load_static(static, new)
    assert torch.equal(static["transformed_image2"], stacked[n:] + 100)
    # two separate tensors -> the stacked static buffer, through its halves
    sep = {"transformed_image1": torch.full((n, 1, 2, 2), 7.0), "transformed_image2": torch.full((n, 1, 2, 2), 9.0),
           "angle_1": torch.ones(n, dtype=torch.float64)}
    Trainer._load_static(static, sep)
    assert float(static["transformed_images"][:n].mean()) == 7.0 and float(static["transformed_images"][n:].mean()) == 9.0
    assert float(static["angle_1"].sum()) == n
    # a static batch WITHOUT the stacked tensor accepts a stacked batch through its halves
    plain = Trainer._clone_batch(sep)
    assert "transformed_images" not in plain
    Trainer._load_static(plain, new)
    assert torch.equal(plain["transformed_image1"], new["transformed_image1"])
    assert torch.equal(SimCLR._two_views(plain), torch.cat([new["transformed_image1"], new["transformed_image2"]]))


@pytest.mark.parametrize("name,depths,widths,kind", [("resnet18", [2, 2, 2, 2], [64, 128, 256, 512], "basic"),
                                                      ("resnet50", [3, 4, 6, 3], [256, 512, 1024, 2048], "bottleneck")])
def test_resnet_arithmetic_matches_an_independent_implementation(name, depths, widths, kind):
    """torchvision (the reference's encoder, resnet_model.py:15,31-43) is not installed here, but `transformers`
    ships an independent implementation of the same published architecture (its ResNetModel hosts the converted
    torchvision checkpoints: v1.5 bottlenecks with the stride on the 3x3, conv7x7/2 + max-pool stem, 1x1-conv + BN
    shortcuts).  With the in-tree network's weights copied into it block by block, both must compute the same
    features -- in eval mode (running statistics) and in train mode (batch statistics, running-stat updates)."""
    transformers = pytest.importorskip("transformers")
    from transformers import ResNetConfig
    from transformers import ResNetModel as HFResNet

    from peclr_amd import resnet

    torch.manual_seed(3)
    mine = getattr(resnet, name)().double()
    with torch.no_grad():          # non-trivial normalisation state everywhere
        for m in mine.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.3, 0.3)
                m.running_var.uniform_(0.5, 2.0)
    hf = HFResNet(ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=widths, depths=depths, layer_type=kind,
                               hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)).double()

    def put(conv_layer, conv, bn):
        conv_layer.convolution.load_state_dict(conv.state_dict())
        conv_layer.normalization.load_state_dict(bn.state_dict())

    put(hf.embedder.embedder, mine.conv1, mine.bn1)
    for s, stage in enumerate((mine.layer1, mine.layer2, mine.layer3, mine.layer4)):
        for b, block in enumerate(stage):
            tgt = hf.encoder.stages[s].layers[b]
            convs = [(block.conv1, block.bn1), (block.conv2, block.bn2)] + ([(block.conv3, block.bn3)] if kind == "bottleneck" else [])
            assert len(tgt.layer) == len(convs)
            for layer, (c, n) in zip(tgt.layer, convs):
                assert layer.convolution.stride == c.stride and layer.convolution.kernel_size == c.kernel_size
                put(layer, c, n)
            if block.downsample is not None:
                put(tgt.shortcut, block.downsample[0], block.downsample[1])
            else:
                assert isinstance(tgt.shortcut, torch.nn.Identity)
    assert sum(p.numel() for p in hf.parameters()) == sum(p.numel() for n_, p in mine.named_parameters() if not n_.startswith("fc."))

    x = torch.randn(3, 3, 64, 64, dtype=torch.float64)

    def features(net):
        return torch.flatten(net.avgpool(net.layer4(net.layer3(net.layer2(net.layer1(net.maxpool(net.relu(net.bn1(net.conv1(x))))))))), 1)

    for training in (False, True):
        mine.train(training)
        hf.train(training)
        want = hf(x).pooler_output.flatten(1)
        got = features(mine)
        assert got.shape == want.shape == (3, widths[-1])
        assert float((got - want).abs().max()) <= 1e-10 * max(1.0, float(want.abs().max())), training
    # train mode moved both sets of running statistics identically (momentum 0.1, unbiased variance)
    last_mine = mine.layer4[-1].bn3 if kind == "bottleneck" else mine.layer4[-1].bn2
    last_hf = hf.encoder.stages[3].layers[-1].layer[-1].normalization
    assert torch.allclose(last_mine.running_var, last_hf.running_var, rtol=1e-12, atol=0)
    assert int(last_mine.num_batches_tracked) == int(last_hf.num_batches_tracked) == 1


def test_device_loss_scaler_is_hip_only_and_cpu_runs_keep_torch_grad_scaler():
    """The device-side scaler lives in HIP memory and inside the fused optimiser; without them precision=16 keeps
    torch's own GradScaler around the foreach optimiser (test above) -- nothing silently degrades."""
    from peclr_amd import _capi
    from peclr_amd.optim import DeviceLossScaler, LARSAdam

    with pytest.raises(_capi.PeclrHipError, match="HIP device memory"):
        DeviceLossScaler("cpu")
    with pytest.raises(ValueError):
        DeviceLossScaler("cpu", growth_factor=1.0)
    opt = LARSAdam([torch.nn.Parameter(torch.ones(3))], lr=1e-3, fused=False)
    with pytest.raises(_capi.PeclrHipError, match="fused HIP optimiser"):
        opt.attach_scaler(object())


def test_x6_routing_rules_and_cpu_fallback(monkeypatch):
    """Which fp32 1x1 products go to peclr_gemm_x6_f32 / _tn_f32 (measured thresholds, bn2d._x6_pays /
    _x6_wgrad_pays), and that a bn2d.Conv2d with hip_gemm set is the stock convolution on tensors the HIP path does
    not take (CPU, NCHW, strided, 3x3)."""
    from peclr_amd import bn2d as B

    r56, r28, r14, r7 = 256 * 56 * 56, 256 * 28 * 28, 256 * 14 * 14, 256 * 7 * 7
    # forward y[R, Cout] = x[R, Cin] W^T: (rows, n_out = Cout, k = Cin)
    assert B._x6_pays(r28, 128, 512) and B._x6_pays(r14, 256, 1024) and B._x6_pays(r14, 1024, 256) and B._x6_pays(r7, 2048, 512)
    # layer1 (64 <-> 256 channels at 8e5 rows): in-tree since round 3 -- the GEMM is a draw, the fused BatchNorm statistics /
    # backward reduction are the gain; the same widths at few rows stay on MIOpen
    assert B._x6_pays(r56, 64, 256) and B._x6_pays(r56, 256, 64) and B._x6_pays(r56, 64, 64) and B._x6_pays(r28, 512, 128)
    assert not B._x6_pays(48 * 56 * 56, 64, 256) and not B._x6_pays(r56, 256, 24) and not B._x6_pays(r56, 32, 256)
    assert not B._x6_pays(16 * 14 * 14, 1024, 256)                                                        # too few tiles
    assert B._x6_wgrad_pays(r28, 128, 512) and B._x6_wgrad_pays(r7, 512, 2048)
    assert B._x6_wgrad_pays(r56, 64, 256) and B._x6_wgrad_pays(r56, 256, 64)      # layer1: the 64-wide tiles of peclr_gemm_x6t_f32
    assert not B._x6_wgrad_pays(r56, 32, 256) and not B._x6_wgrad_pays(4096, 256, 1024)
    with B.routing(x6_layer1_wgrad=False):
        assert not B._x6_wgrad_pays(r56, 64, 256) and not B._x6_wgrad_pays(r56, 256, 64) and B._x6_wgrad_pays(r28, 128, 512)
    assert B.ROUTING.x6_layer1_wgrad
    # the thresholds are roofline statements, not ResNet-50 @224's row counts in disguise: the 64-channel rule is "operands +
    # output exceed the Infinity Cache" (the fused BatchNorm pass would otherwise come from HBM again) ...
    assert B.ROUTING.streams_from_hbm(r56, 64, 64) and not B.ROUTING.streams_from_hbm(64 * 56 * 56, 64, 256)
    assert B._x6_pays(128 * 112 * 112, 64, 256) and B._x6_pays(128 * 112 * 112, 256, 64)        # C5's layer1: 2 x 64 views @448
    # ... and `force` routes everything the kernels accept (tests exercise the full-size composition on small problems)
    with B.routing(force=True):
        assert B._x6_pays(16 * 14 * 14, 1024, 256) and B._x6_pays(48 * 56 * 56, 64, 256) and B._x6_wgrad_pays(4096, 256, 1024)
        assert not B._x6_pays(r56, 256, 24) and not B._x6_pays(r56, 32, 256)                       # the kernels' own shape limits stay
    with pytest.raises(AttributeError):
        with B.routing(no_such_switch=1):
            pass
    torch.manual_seed(0)
    conv = B.Conv2d(256, 512, 1, bias=False)
    ref = torch.nn.Conv2d(256, 512, 1, bias=False)
    ref.load_state_dict(conv.state_dict())
    conv.hip_gemm = True
    x = torch.randn(2, 256, 5, 5)
    for inp in (x, x.contiguous(memory_format=torch.channels_last)):
        a, b = inp.clone().requires_grad_(), inp.clone().requires_grad_()
        ya, yb = conv(a), ref(b)
        assert torch.equal(ya, yb)
        ya.sum().backward()
        yb.sum().backward()
        assert torch.equal(a.grad, b.grad) and torch.equal(conv.weight.grad, ref.weight.grad)
        conv.weight.grad = ref.weight.grad = None


def test_compact_shortcut_gradient_protocol():
    """bn2d._compact_grad / _take_compact / _expand_compact: the zero-stride NaN view a 1x1 / stride-2 shortcut returns as its
    input gradient carries the compact [N, H/2, W/2, C] gradient to the entry-gradient GEMM exactly once; anything that is
    not such a view is left alone, and the dense fallback puts the compact values at the even pixels."""
    from peclr_amd import bn2d as B

    dc = torch.arange(2 * 2 * 3 * 5, dtype=torch.float32).view(2 * 2 * 3, 5)          # N=2, H/2=2, W/2=3, C=5
    shape = (2, 5, 4, 6)
    s1 = B._compact_grad(dc, shape)
    s2 = B._compact_grad(dc + 1, shape)
    assert tuple(s1.shape) == shape and not any(s1.stride()) and torch.isnan(s1).all()     # loud if ever added to anything
    assert s1.data_ptr() != s2.data_ptr()
    assert B._take_compact(torch.zeros(shape)) is None and B._take_compact(None) is None
    assert B._take_compact(s1.expand(shape)[:, :3]) is None                                 # another shape: not ours
    got = B._take_compact(s2)
    assert torch.equal(got, dc + 1) and B._take_compact(s2) is None                         # consumed
    full = B._expand_compact(B._take_compact(s1), shape)
    assert not B._COMPACT and full.is_contiguous(memory_format=torch.channels_last)
    ref = torch.zeros(shape)
    ref[:, :, ::2, ::2] = dc.view(2, 2, 3, 5).permute(0, 3, 1, 2)
    assert torch.equal(full, ref)
    # the (dy, mask) form of an identity shortcut's gradient: its dense equivalent is dy where the mask bit is set
    dy = torch.randn(2, 64, 3, 3).contiguous(memory_format=torch.channels_last)
    bits = torch.rand(2 * 9, 64) > 0.5
    words = (bits.view(18, 2, 32).to(torch.int64) << torch.arange(32)).sum(-1)
    mask = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
    s3 = B._lazy_grad(("mask", dy, mask), dy.shape, dy.device)
    dense = B._take_compact(s3)
    assert torch.equal(dense, dy * bits.view(2, 3, 3, 64).permute(0, 3, 1, 2)) and not B._COMPACT


def test_capture_guard_holds_the_collector_and_restores_it(monkeypatch):
    """Trainer._capturing (hipGraph lifetime): garbage is collected before the capture starts, the
    cyclic collector is off while it records (also when the recorded code raises) and back afterwards; the graph object is
    kept until close().  The capture itself is replaced by a recording stand-in (no GPU here)."""
    import contextlib
    import gc

    from peclr_amd import Trainer

    events = []

    @contextlib.contextmanager
    def fake_graph(graph, **kw):
        events.append(("enter", gc.isenabled(), kw))
        yield
        events.append(("exit", gc.isenabled()))

    monkeypatch.setattr(torch.cuda, "graph", fake_graph)
    collected = []
    monkeypatch.setattr(gc, "collect", lambda *a: collected.append(len(events)) or 0)
    tr = Trainer()
    g1, g2 = object(), object()
    assert gc.isenabled()
    with tr._capturing(g1, pool="p"):
        assert not gc.isenabled()
    assert gc.isenabled() and tr._graphs_alive == [g1]
    assert events == [("enter", False, {"pool": "p"}), ("exit", False)] and collected == [0]   # collected BEFORE entering
    with pytest.raises(RuntimeError):
        with tr._capturing(g2):
            raise RuntimeError("the recorded step failed")
    assert gc.isenabled() and tr._graphs_alive == [g1]           # a failed capture is not kept
    gc.disable()
    try:
        with tr._capturing(g2):
            pass
        assert not gc.isenabled()                                 # the caller's own setting is respected
    finally:
        gc.enable()


def test_checkpoint_shifts_belong_to_the_invocation_not_to_the_layer():
    """Advisor, round 4: the copy of the running mean a checkpointed block's first run centres its statistics on used to be
    ONE slot per BatchNorm module, overwritten by every first run -- two checkpointed forward passes before one backward (two
    view passes, a no-grad forward in between) made the first pass's re-run centre on the later mean.  It now lives on the
    invocation's autograd context: every re-run reads back exactly what ITS first run took."""
    from peclr_amd import bn2d as B

    class FakeBN:
        def __init__(self):
            self.running_mean = torch.zeros(3)

    bn = FakeBN()
    seen = []

    def run(x):
        shift = B._ckpt_shift_of(bn)
        seen.append(("first" if B._FIRST_RUN else "rerun", None if shift is None else float(shift[0])))
        if B._FIRST_RUN:
            bn.running_mean += 1.0              # what the statistics update does between the two passes
        return x * 2.0

    assert B._ckpt_shift_of(bn) is None          # outside a checkpointed block: nothing is kept
    x1 = torch.ones(2, requires_grad=True)
    x2 = torch.ones(2, requires_grad=True)
    y1 = B.checkpoint_block(run, x1)             # first run #1 centres on 0
    with torch.no_grad():
        B.checkpoint_block(run, x1)              # a no-grad pass: plain, keeps nothing
    y2 = B.checkpoint_block(run, x2)             # first run #2 centres on 2 (two updates so far)
    (y1.sum() + y2.sum()).backward()             # re-runs in reverse order
    assert seen == [("first", 0.0), ("rerun", None), ("first", 1.0), ("rerun", 1.0), ("rerun", 0.0)], seen
    assert torch.equal(x1.grad, torch.full((2,), 2.0)) and B._CKPT_SHIFTS is None
    assert not hasattr(bn, "_ckpt_shift") and not hasattr(bn, "_sync_shift")   # nothing parked on the module (deepcopy / pickle / .to())


def test_side_channels_are_drained_by_the_autograd_engine_when_a_backward_pass_ends():
    """Advisor, round 4: `_BN_BWD_STATS` / `_COMPACT` were emptied only by Trainer._join_wgrad and by a test fixture; any other
    training loop left stale entries pinning device tensors.  Their producers now queue `end_backward` on the engine."""
    from peclr_amd import bn2d as B

    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            B._BN_BWD_STATS[12345] = ("token", None, 0, 0)       # what _note_bn_bwd parks: nobody will pop this one
            B._drain_after_backward()
            assert len(B._BN_BWD_STATS) == 1                      # still there while the pass runs
            return g * 2.0

    B.end_backward()
    x = torch.ones(3, requires_grad=True)
    Producer.apply(x).sum().backward()                            # a plain loop: no Trainer, no fixture
    assert B._BN_BWD_STATS == {} and B._COMPACT == {} and B.last_backward_leftovers == 1 and not B._DRAIN_QUEUED
    Producer.apply(x).sum().backward()
    assert B.last_backward_leftovers == 1
    B._drain_after_backward()                                     # outside a backward pass: nothing to queue on, no error
    assert not B._DRAIN_QUEUED


def test_deferred_batchnorm_placeholder_and_who_may_take_it():
    """The placeholder a BatchNorm layer returns when it leaves its apply pass to its consumer: NaN under every index, one element,
    zero strides, never the address of a lazy-gradient sentinel; on a CPU tensor no consumer takes the hand-over (the layer then
    writes its output as always)."""
    import torch
    from peclr_amd import bn2d as B

    x = torch.zeros(2, 8, 4, 5)
    v = B._deferred_view(x)
    assert v.shape == x.shape and v.dtype == x.dtype and not any(v.stride()) and bool(torch.isnan(v).all())
    g = B._lazy_grad(("s2", torch.zeros(1)), (2, 8, 4, 5), x.device)
    try:
        assert g.data_ptr() != v.data_ptr()
    finally:
        B._COMPACT.clear()
    conv = B.Conv2d(8, 16, 1, bias=False)
    bn = B.FusedBatchNormAct2d(8)
    assert not hasattr(conv, "_takes_deferred") and not bn._takes_deferred_residual(x)
    with B.routing(bn_shortcut_in_add=True):
        assert not bn._takes_deferred_residual(x)      # (no HIP tensors, layers not switched to the HIP kernels)
    y = bn(x, None, True, consumer=conv)                       # stock path: a real tensor, no placeholder
    assert not hasattr(y, "_peclr_deferred") and not torch.isnan(y).any()


def test_absmax_tag_is_dropped_when_the_tensor_is_written_after_the_pass_that_measured_it():
    """The "pair" GEMMs scale an operand into fp16 range by the maximum its producing pass left beside it; a later in-place write
    would make that maximum stale (overflow to inf in the worst case), so the tag carries the tensor's version counter and an
    out-of-date tag reads as "no maximum known" -- the launch then takes the six-product kernel, which needs none."""
    from peclr_amd import bn2d

    with bn2d.routing(x6_pair=True):
        t = torch.ones(4, 4)
        slot = torch.ones(1)
        assert bn2d._absmax_of(t) is None
        bn2d._tag_absmax(t, slot)
        assert bn2d._absmax_of(t) is slot
        assert bn2d._absmax_of(t.view(16)) is None          # the tag belongs to the tensor object, not to its storage
        t.mul_(3.0)
        assert bn2d._absmax_of(t) is None
        bn2d._tag_absmax(t, slot)
        assert bn2d._absmax_of(t) is slot
    with bn2d.routing(x6_pair=False):
        assert bn2d._absmax_of(t) is None
