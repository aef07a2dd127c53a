"""precision=16 (the reference's default, training_config.json:9; Lightning 1.0.8 native AMP = fp16 autocast +
torch.cuda.amp.GradScaler around the optimiser, peclr_training.py:78-79) with the loss scaler folded into the
fused HIP optimiser step: `DeviceLossScaler` + peclr_lars_sumsq_amp_f32 / peclr_lars_adam_update_amp_f32 /
peclr_amp_update.  The comparison arm is torch.amp.GradScaler itself driving the same fused optimiser."""
import copy
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _param_sets():
    g = torch.Generator().manual_seed(7)
    shapes = [(64, 3, 7, 7), (64,), (256, 64, 1, 1), (5000,), (33, 17), (1,), (4096,), (4097,)]
    vals = [torch.randn(*s, generator=g) * 0.1 for s in shapes]
    vals[1] = torch.zeros(64)                                  # |p| == 0: LARS leaves that gradient alone

    def make():
        ps = [torch.nn.Parameter(v.clone().to(DEV)) for v in vals]
        ps[0] = torch.nn.Parameter(vals[0].clone().to(DEV).contiguous(memory_format=torch.channels_last))
        return ps

    return shapes, make


@pytest.mark.parametrize("lars,write_back", [(True, False), (True, True), (False, False)], ids=["lars", "lars_wb", "adam"])
def test_device_loss_scaler_matches_torch_grad_scaler(lars, write_back):
    """Same gradients (scaled, with inf / nan injected at some steps), two arms: torch's GradScaler.step/update
    around the fused optimiser, and the device-side scaler inside it.  Same parameters, moments, scale, growth
    tracker and Adam step count after every step."""
    from peclr_amd.optim import DeviceLossScaler, LARSAdam

    shapes, make = _param_sets()
    pa, pb = make(), make()

    def groups(ps):
        return [{"params": [p for p in ps if p.dim() > 1], "weight_decay": 1e-3},
                {"params": [p for p in ps if p.dim() <= 1], "weight_decay": 0.0}]

    oa = LARSAdam(groups(pa), lr=1e-2, lars=lars, write_back=write_back, fused=True)
    ob = LARSAdam(groups(pb), lr=1e-2, lars=lars, write_back=write_back, fused=True)
    sa = DeviceLossScaler(DEV, init_scale=2.0 ** 10, growth_interval=3)
    oa.attach_scaler(sa)
    sb = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10, growth_interval=3)
    g = torch.Generator().manual_seed(8)
    bad_steps = {4: float("inf"), 5: float("nan"), 10: float("-inf")}
    taken = 0
    for step in range(14):
        scale = sa.get_scale()
        assert scale == sb.get_scale()
        for grp_a, grp_b in zip(oa.param_groups, ob.param_groups):
            grp_a["lr"] = grp_b["lr"] = 1e-2 * (1 + step) / 14
        for i, (a, b, s) in enumerate(zip(pa, pb, shapes)):
            gr = (torch.randn(*s, generator=g) * scale).to(DEV)
            if step in bad_steps and i == (step % len(pa)):
                gr.view(-1)[gr.numel() // 2] = bad_steps[step]
            a.grad = gr.clone().contiguous(memory_format=torch.channels_last) if a.dim() == 4 else gr.clone()
            b.grad = gr.clone().contiguous(memory_format=torch.channels_last) if b.dim() == 4 else gr.clone()
        before = [p.detach().clone() for p in pa]
        oa.step()
        sb.scale(torch.ones(1, device=DEV))          # lazily creates torch's device-side scale
        sb.step(ob)
        sb.update()
        if step in bad_steps:
            assert all(torch.equal(p, q) for p, q in zip(pa, before)), "a step with inf / nan gradients must be skipped"
        else:
            taken += 1
            assert any(not torch.equal(p, q) for p, q in zip(pa, before))
        for i, (a, b) in enumerate(zip(pa, pb)):
            # bias corrections: pow() in double on the device vs `**` on the host, rounded to fp32 -- equal up to the
            # last bit of that rounding (the updates are ~lr = 1e-3..1e-2 here)
            assert torch.allclose(a.detach(), b.detach(), rtol=1e-6, atol=3e-9), (step, i, float((a - b).abs().max()))
            for key in ("exp_avg", "exp_avg_sq"):
                if key in oa.state[a]:
                    # the two template instantiations may contract b*m + (1-b)*g into different FMAs: last-bit
                    # differences of the TERMS, which show where they cancel
                    ref = ob.state[b][key]
                    assert torch.allclose(oa.state[a][key], ref, rtol=1e-6, atol=1e-6 * float(ref.abs().max())), (step, i, key)
            if write_back:                            # what the reference's wrapper leaves in p.grad (unscaled; LARS-scaled if stepped)
                assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-9, equal_nan=True), (step, i)
    da, db = sa.state_dict(), sb.state_dict()
    assert da == db, (da, db)
    assert sa.good_steps() == taken == 11
    steps_a = {int(st["step"]) for st in oa.state_dict()["state"].values()}
    steps_b = {int(st["step"]) for st in ob.state_dict()["state"].values()}
    assert steps_a == steps_b == {taken}
    # either class loads the other's state
    sb2 = torch.amp.GradScaler("cuda")
    sb2.load_state_dict(da)
    assert sb2.get_scale() == sa.get_scale()
    sa2 = DeviceLossScaler(DEV)
    sa2.load_state_dict(db)
    assert sa2.state_dict() == db


def test_scale_growth_stops_at_the_largest_finite_scale():
    from peclr_amd import _capi
    from peclr_amd.optim import DeviceLossScaler

    s = DeviceLossScaler(DEV, init_scale=2.0 ** 127, growth_interval=1)
    _capi._check(_capi.lib().peclr_amp_update(s.state.data_ptr(), 2.0, 0.5, 1, torch.cuda.current_stream().cuda_stream),
                 "peclr_amp_update")
    assert s.get_scale() == 2.0 ** 127 and s.good_steps() == 1      # 2^128 is not finite in fp32: torch keeps the scale
    assert _capi.lib().peclr_amp_update(None, 2.0, 0.5, 1, None) == -1      # PECLR_ERR_NULL
    assert _capi.lib().peclr_amp_update(s.state.data_ptr(), 1.0, 0.5, 1, None) == -2   # PECLR_ERR_SHAPE
    with pytest.raises(_capi.PeclrHipError):
        DeviceLossScaler("cpu")


def _scalers_agree(a, b):
    """Two runs of the same model: MIOpen's atomically accumulated weight gradients differ in the last bits from run
    to run, so a gradient that sits exactly at the fp16 overflow threshold may skip in one run and not in the other.
    Same algorithm => at most one skip apart (the bit-exact comparison against torch's GradScaler is the first test)."""
    sa, sb = a.state_dict(), b.state_dict()
    return (abs(np.log2(sa["scale"] / sb["scale"])) <= 1.0 and abs(a.good_steps() - b.good_steps()) <= 1
            and {k: sa[k] for k in ("growth_factor", "backoff_factor", "growth_interval")} ==
                {k: sb[k] for k in ("growth_factor", "backoff_factor", "growth_interval")})


def _model_and_batch(seed, n=8, resnet="18", din=512, accum=1, size=64):
    from peclr_amd import Hybrid2Model, hybrid2_config
    from peclr_amd.bn2d import enable_hip_batchnorm

    warnings.simplefilter("ignore")
    torch.manual_seed(seed)
    cfg = hybrid2_config(resnet_size=resnet, projection_head_input_dim=din, augmentation=["crop", "rotate"],
                         batch_size=n, num_samples=64, warmup_epochs=1, num_of_mini_batch=accum, pretrained=False)
    base = Hybrid2Model(cfg).to(DEV).train()
    base.encoder = base.encoder.to(memory_format=torch.channels_last)
    enable_hip_batchnorm(base.encoder)
    return base, _batch(seed + 1, n, size)


def _batch(seed, n, size=64):
    g = torch.Generator().manual_seed(seed)
    b = {"transformed_image1": torch.randn(n, 3, size, size, generator=g), "transformed_image2": torch.randn(n, 3, size, size, generator=g),
         "jitter_x_1": torch.randint(-14, 1, (n,), generator=g), "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
         "jitter_y_1": torch.randint(-14, 1, (n,), generator=g), "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
         "angle_1": torch.randint(-45, 46, (n,), generator=g).double(), "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}
    b = {k: v.to(DEV) for k, v in b.items()}
    for k in ("transformed_image1", "transformed_image2"):
        b[k] = b[k].contiguous(memory_format=torch.channels_last)
    return b


def test_eager_precision_16_equals_torch_grad_scaler_loop():
    """The Trainer's fp16 step (device-side scaler) against the textbook loop: autocast, GradScaler.scale(loss)
    .backward(), GradScaler.step(optimizer), GradScaler.update() on a copy of the model."""
    from peclr_amd import Trainer
    from peclr_amd.optim import DeviceLossScaler

    base, batch = _model_and_batch(71)
    ma, mb = copy.deepcopy(base), copy.deepcopy(base)
    ta = Trainer(max_epochs=10, precision=16).attach(ma)
    tb = Trainer(max_epochs=10, precision="fp32").attach(mb)     # only to build optimiser + schedule the same way
    ta.zero_grad()
    tb.zero_grad()
    sb = torch.amp.GradScaler("cuda")
    la, lb, scales = [], [], []
    for i in range(10):
        la.append(float(ta.training_micro_step(batch, i)["loss"]))
        with torch.autocast("cuda", dtype=torch.float16):
            out = mb.training_step(batch, i)
        sb.scale(out["loss"]).backward()
        sb.step(tb.optimizer)
        sb.update()
        tb.optimizer.zero_grad(set_to_none=True)
        tb.scheduler.step()
        lb.append(float(out["loss"]))
        scales.append((ta._scaler.get_scale(), sb.get_scale()))
    assert isinstance(ta._scaler, DeviceLossScaler)
    assert all(abs(np.log2(a / b)) <= 1.0 for a, b in scales), scales   # the same steps overflowed and were skipped
    assert la == pytest.approx(lb, rel=3e-2)                       # MIOpen's atomic weight gradients, nothing else
    assert la[-1] < la[0]
    assert abs(ta._scaler.good_steps() - int(next(iter(tb.optimizer.state.values()))["step"])) <= 1


def test_whole_step_graph_in_precision_16_matches_eager_and_skips_on_the_device():
    """forward (fp16 autocast) + scaled backward + unscale / inf check / skip / update / scale update in ONE
    hipGraph.  Same curve as the eager fp16 loop; an overflow forced between two replays is skipped by the replay
    itself (weights and moments untouched, scale halved, Adam's step count not advanced) with no host involvement."""
    from peclr_amd import Trainer

    base, batch = _model_and_batch(73)
    steps = 12
    te = Trainer(max_epochs=10, precision=16).attach(copy.deepcopy(base))
    te.zero_grad()
    eager = [float(te.training_micro_step(batch, i)["loss"]) for i in range(steps)]
    mg = copy.deepcopy(base)
    tg = Trainer(max_epochs=10, precision=16).attach(mg)
    tg.zero_grad()
    tg.capture_step_graph(batch, warmup=3)
    graph = [float(tg.replay_step()["loss"]) for _ in range(steps - 4)]
    assert tg.global_step == te.global_step == steps
    assert _scalers_agree(tg._scaler, te._scaler) and tg._scaler.good_steps() <= steps
    assert graph == pytest.approx(eager[4:], rel=6e-2)
    assert graph[-1] < eager[0]
    # force an overflow: 2^100 * loss is inf in the fp16 backward
    torch.cuda.synchronize()
    good, scale = tg._scaler.good_steps(), 2.0 ** 100
    tg._scaler._f[0] = scale
    w = [p.detach().clone() for p in mg.parameters()]
    m = [tg.optimizer.state[p]["exp_avg"].clone() for p in mg.parameters() if p in tg.optimizer.state]
    for k in range(3):
        out = tg.replay_step()
        assert np.isfinite(float(out["loss"]))                     # the reported loss is the unscaled one
        assert tg._scaler.get_scale() == scale / 2 ** (k + 1)
    assert all(torch.equal(a, b) for a, b in zip(w, mg.parameters()))
    assert all(torch.equal(a, tg.optimizer.state[p]["exp_avg"]) for a, p in
               zip(m, [p for p in mg.parameters() if p in tg.optimizer.state]))
    assert tg._scaler.good_steps() == good and tg.global_step == steps + 3   # the LR schedule advances regardless
    tg._scaler._f[0] = 1024.0
    tg.replay_step()
    assert tg._scaler.good_steps() == good + 1
    assert any(not torch.equal(a, b) for a, b in zip(w, mg.parameters()))
    assert int(tg.optimizer.state_dict()["state"][0]["step"]) == good + 1    # checkpoints carry the device's count


def test_split_graphs_in_precision_16_match_eager():
    from peclr_amd import Trainer

    base, batch = _model_and_batch(75)
    steps = 10
    te = Trainer(max_epochs=10, precision=16, grad_buckets=True).attach(copy.deepcopy(base))
    te.zero_grad()
    eager = [float(te.training_micro_step(batch, i)["loss"]) for i in range(steps)]
    tg = Trainer(max_epochs=10, precision=16, grad_buckets=True).attach(copy.deepcopy(base))
    tg.zero_grad()
    tg.capture_split_graphs(batch, warmup=2)
    graph = [float(tg.replay_split()["loss"]) for _ in range(steps - 2)]
    assert tg.global_step == te.global_step == steps
    assert _scalers_agree(tg._scaler, te._scaler)
    assert graph == pytest.approx(eager[2:], rel=6e-2)
    assert graph[-1] < eager[0]


def test_micro_batch_graph_with_accumulation_in_precision_16():
    from peclr_amd import Trainer

    base, batch = _model_and_batch(77, accum=2)
    micro = 16
    te = Trainer(max_epochs=10, precision=16, accumulate_grad_batches=2).attach(copy.deepcopy(base))
    te.zero_grad()
    eager = [float(te.training_micro_step(batch, i)["loss"]) for i in range(micro)]
    tg = Trainer(max_epochs=10, precision=16, accumulate_grad_batches=2).attach(copy.deepcopy(base))
    tg.zero_grad()
    tg.capture_micro_graph(batch, warmup_windows=1)
    graph = [float(tg.replay_micro()["loss"]) for _ in range(micro - 2)]
    assert tg.global_step == te.global_step == micro // 2
    assert _scalers_agree(tg._scaler, te._scaler)
    assert graph == pytest.approx(eager[2:], rel=6e-2)
    assert graph[-1] < eager[0]


def test_fit_in_precision_16_with_hip_graph_checkpoints_the_scaler(tmp_path):
    """fit(hip_graph=True, precision=16): replays for equal shapes, eager for the ragged tail, and a checkpoint
    whose `native_amp_scaling_state` torch's own GradScaler loads and whose optimiser step count is the number
    of steps actually taken; resuming restores both."""
    from peclr_amd import Trainer

    base, _ = _model_and_batch(79)

    def batches(epoch):
        for j, n in enumerate((8, 8, 8, 8, 6)):
            yield _batch(1000 + 10 * epoch + j, n)

    runs = {}
    for graph in (False, True):
        m = copy.deepcopy(base)
        tr = Trainer(max_epochs=2, checkpoint_dir=str(tmp_path / f"ck{int(graph)}"), hip_graph=graph, precision=16)
        os.makedirs(tr.checkpoint_dir, exist_ok=True)
        tr.fit(m, batches)
        torch.cuda.synchronize()
        runs[graph] = (tr, m)
    (te, me), (tg, mg) = runs[False], runs[True]
    assert tg.global_step == te.global_step == 10
    assert _scalers_agree(tg._scaler, te._scaler)
    assert float(mg.train_metrics_epoch["loss"]) == pytest.approx(float(me.train_metrics_epoch["loss"]), rel=6e-2)
    (name,) = os.listdir(tg.checkpoint_dir)
    ckpt = torch.load(os.path.join(tg.checkpoint_dir, name), map_location="cpu")
    torch.amp.GradScaler("cuda").load_state_dict(ckpt["native_amp_scaling_state"])
    assert ckpt["native_amp_scaling_state"]["scale"] == tg._scaler.get_scale()
    assert {int(st["step"]) for st in ckpt["optimizer_states"][0]["state"].values()} == {tg._scaler.good_steps()}
    tr2 = Trainer(max_epochs=3, precision=16).attach(copy.deepcopy(base)).resume(os.path.join(tg.checkpoint_dir, name))
    assert tr2._scaler.state_dict() == tg._scaler.state_dict() and tr2._scaler.good_steps() == tg._scaler.good_steps()
