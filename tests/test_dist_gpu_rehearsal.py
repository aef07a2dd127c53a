"""N > 1 on the REAL kernels with one GPU: two processes share cuda:0 and talk over gloo (RCCL refuses
two ranks per device).  Everything except the transport is the production multi-GPU path: rank offsets
in the NT-Xent kernels, the packed lse/loss gather, gradients accumulated into the flat buckets with
stride-preserving views, SUM all-reduce from autograd hooks, the fused optimiser on bucket views."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys, warnings
sys.path.insert(0, os.environ["PECLR_ROOT"])
warnings.simplefilter("ignore")
import torch
# Run-to-run repeatable arms: every shape the in-tree kernels accept goes to them (PECLR_ROUTE_FORCE=1 in the environment: fixed-order
# slab weight gradients), and whatever is left on MIOpen must pick a deterministic algorithm (no atomically accumulated split-K).
torch.use_deterministic_algorithms(True, warn_only=True)
torch.backends.cudnn.deterministic = True
from peclr_amd import Hybrid2Model, Trainer, hybrid2_config
from peclr_amd import dist as pdist
from peclr_amd.bn2d import enable_hip_batchnorm

N_LOCAL = 4
SYNC_BN = os.environ.get("PECLR_SYNC_BN", "0") == "1"   # train-mode BN with global-batch statistics
def make_model():
    torch.manual_seed(21)
    cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, augmentation=["crop", "rotate"],
                         batch_size=N_LOCAL, num_samples=64, pretrained=False)
    m = Hybrid2Model(cfg).cuda().train()
    m.encoder = m.encoder.to(memory_format=torch.channels_last)
    enable_hip_batchnorm(m.encoder)
    if not SYNC_BN:
        for mod in m.modules():      # batch-independent normalisation: isolates the collective logic
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.eval()
    return m
def make_batch(world):
    g = torch.Generator().manual_seed(7)
    n = world * N_LOCAL
    b = {"transformed_image1": torch.randn(n, 3, 32, 32, generator=g), "transformed_image2": torch.randn(n, 3, 32, 32, generator=g),
         "jitter_x_1": torch.randint(-14, 1, (n,), generator=g), "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
         "jitter_y_1": torch.randint(-14, 1, (n,), generator=g), "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
         "angle_1": torch.randint(-45, 46, (n,), generator=g).double(), "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}
    return b
def to_dev(b):
    b = {k: v.cuda() for k, v in b.items()}
    for k in ("transformed_image1", "transformed_image2"):
        b[k] = b[k].contiguous(memory_format=torch.channels_last)
    return b
world = int(os.environ.get("WORLD_SIZE", "1"))
pdist.init_from_env()
rank = pdist.rank()
model = make_model()
tr = Trainer(max_epochs=1, bucket_bytes=4 << 20, sync_batchnorm=SYNC_BN).attach(model)
tr.zero_grad()
full = make_batch(max(world, 2))
if world > 1:
    sl = slice(rank * N_LOCAL, (rank + 1) * N_LOCAL)
    batch = to_dev({k: v[sl].contiguous() for k, v in full.items()})
    tr.reducer.prepare(tr._unused)
else:
    batch = to_dev(full)
out = model.training_step(batch, 0)
out["loss"].backward()
if world > 1:
    tr.reducer.finish()
grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None and "final_layer" not in n}
bufs = {n: b.detach().double().cpu() for n, b in model.named_buffers()}
torch.save({"loss": out["loss"].detach().cpu(), "grads": grads, "buffers": bufs}, os.environ["PECLR_OUT"] + f".r{rank}")
tr.optimizer.step()                      # fused LARS/Adam straight out of the flat buckets
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
"""


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


SPLIT_WORKER = WORKER.split("world = int(os.environ")[0] + r"""
world = int(os.environ.get("WORLD_SIZE", "1"))
pdist.init_from_env()
rank = pdist.rank()
model = make_model()
tr = Trainer(max_epochs=10, bucket_bytes=4 << 20, grad_buckets=True).attach(model)
tr.zero_grad()
full = make_batch(2)
sl = slice(rank * N_LOCAL, (rank + 1) * N_LOCAL) if world > 1 else slice(0, 2 * N_LOCAL)
batch = to_dev({k: v[sl].contiguous() for k, v in full.items()})
losses = []
if os.environ["PECLR_MODE"] == "split":
    tr.capture_split_graphs(batch, warmup=1)
    losses = [float(tr.replay_split()["loss"]) for _ in range(3)]
else:
    losses = [float(tr.training_micro_step(batch, i)["loss"]) for i in range(4)][1:]
torch.cuda.synchronize()
torch.save({"losses": losses, "w": model.projection_head[3].weight.detach().cpu()}, os.environ["PECLR_OUT"] + f".r{rank}")
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
"""


@pytest.mark.timeout(600)
def test_split_graphs_two_ranks_equal_eager_two_ranks(tmp_path):
    """capture_split_graphs / replay_split across two ranks (collectives between the graphs, gradients
    copied from the captured backward into the all-reduce buckets) against the eager two-rank loop."""
    script = tmp_path / "worker.py"
    script.write_text(SPLIT_WORKER)
    env = dict(os.environ, PECLR_ROOT=ROOT, PECLR_DIST_BACKEND="gloo", PECLR_SHARE_DEVICE="1", PECLR_SYNC_BN="0")
    res = {}
    for mode in ("eager", "split"):
        port = free_port()
        procs = [subprocess.Popen([sys.executable, str(script)],
                                  env=dict(env, PECLR_MODE=mode, PECLR_OUT=str(tmp_path / mode), RANK=str(r), LOCAL_RANK=str(r),
                                           WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        outs = [p.communicate(timeout=500)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        res[mode] = [torch.load(str(tmp_path / mode) + f".r{r}") for r in range(2)]
    assert res["split"][0]["losses"] == res["split"][1]["losses"]             # global loss: identical on both ranks
    assert torch.equal(res["split"][0]["w"], res["split"][1]["w"])            # replicas stay in lock-step
    assert res["split"][0]["losses"][0] == pytest.approx(res["eager"][0]["losses"][0], rel=2e-3)
    assert res["split"][0]["losses"] == pytest.approx(res["eager"][0]["losses"], rel=6e-2)
    np.testing.assert_allclose(res["split"][0]["w"].numpy(), res["eager"][0]["w"].numpy(), atol=2e-3)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("sync_bn", [False, True], ids=["frozen_bn", "sync_bn"])
def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path, sync_bn):
    """frozen_bn: eval-mode normalisation, exercises the loss/gradient collectives alone.
    sync_bn: train-mode BatchNorm everywhere with Trainer(sync_batchnorm=True): 2 ranks x 4 pairs must
    reproduce one device with 8 pairs -- loss, every gradient and the running statistics."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    # The shapes of this rehearsal (4 pairs of small images) lie below the in-tree kernels' routing thresholds; left alone most weight
    # gradients would come from MIOpen's atomically accumulated split-K kernels and two runs of the SAME arm would differ.  Both arms
    # therefore run with PECLR_ROUTE_FORCE=1 (in-tree fixed-order weight gradients wherever the kernels accept the shape) and under
    # torch.use_deterministic_algorithms (a deterministic MIOpen algorithm for the rest): each arm repeats itself bit for bit (asserted
    # below by running the one-rank arm twice), so the two-rank result differs from the one-rank result by the reduction trees only,
    # and ONE comparison decides.
    seen = measure_two_ranks_against_one(tmp_path / "run", script, sync_bn)
    bars = BARS[sync_bn]
    misses = {k: (v, bars[k]) for k, v in seen["worst"].items() if not v <= bars[k]}
    assert not misses, f"(observed, bar) {misses}; worst offenders {seen['where']}"


# Bars.  With both arms deterministic the deviation two ranks vs one rank is a CONSTANT of the build (`tools/exp/rehearsal_noise.py`,
# round 6, 10 + 5 runs, every run the same numbers): frozen_bn loss 0, gradients 0 element-wise max, 9.2e-7 norm-wise (one tensor:
# the reduction tree of the summed weight gradient); sync_bn loss 2.4e-7, gradients 4.6e-6 / 4.5e-6, buffers equal.  The bars sit
# ~10x above that and 10 - 4000x below round 5's (loss 2e-6 / 2e-5, gradients 2e-4 / 2e-2, 6e-3 norm-wise, buffers 1e-4).
BARS = {False: {"loss": 5e-7, "grad_max": 1e-5, "grad_norm": 1e-5, "buffer": 0.0},
        True: {"loss": 2.5e-6, "grad_max": 5e-5, "grad_norm": 5e-5, "buffer": 1e-6}}


def measure_two_ranks_against_one(out, script, sync_bn):
    """Runs the two-rank arm once and the one-rank arm twice; asserts everything that must hold EXACTLY (replicas identical, the
    one-rank arm repeats itself bit for bit, processes exit cleanly) and returns the worst observed deviations two ranks vs one."""
    out.mkdir()
    env = dict(os.environ, PECLR_ROOT=ROOT, PECLR_DIST_BACKEND="gloo", PECLR_SHARE_DEVICE="1", PECLR_ROUTE_FORCE="1",
               PECLR_SYNC_BN="1" if sync_bn else "0")
    port = free_port()
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(env, PECLR_OUT=str(out / "two"), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=500)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    for name in ("one", "again"):
        subprocess.run([sys.executable, str(script)], env=dict(env, PECLR_OUT=str(out / name), WORLD_SIZE="1"), check=True, timeout=500)
    r0, r1 = torch.load(str(out / "two") + ".r0"), torch.load(str(out / "two") + ".r1")
    one, again = torch.load(str(out / "one") + ".r0"), torch.load(str(out / "again") + ".r0")
    assert torch.equal(r0["loss"], r1["loss"])                        # global loss: identical on both ranks
    assert torch.equal(one["loss"], again["loss"])                    # the one-rank arm repeats itself bit for bit
    for n, g1 in one["grads"].items():
        assert torch.equal(g1, again["grads"][n]), f"one-rank arm is not run-to-run repeatable: {n}"
    worst = {"loss": abs(float(r0["loss"]) - float(one["loss"])), "grad_max": 0.0, "grad_norm": 0.0, "buffer": 0.0}
    where = {}
    # train-mode BN over few rows (the last stages normalise over 16 rows here) amplifies the fp32 summation-order
    # difference of the two reduction trees; a wrong count / shift / sum would be O(1)
    for n, g1 in one["grads"].items():
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n      # SUM-reduced: identical on both ranks
        if sync_bn and (n.endswith("0.bias") and "projection_head" in n):
            continue  # bias in front of a train-mode BN: gradient is rounding noise around 0 on both sides
        a, b = r0["grads"][n].double().numpy(), g1.double().numpy()
        rel_max = max(0.0, float(np.abs(a - b).max()) - 1e-7) / max(1e-6, float(np.abs(b).max()))
        if rel_max > worst["grad_max"]:
            worst["grad_max"], where["grad_max"] = rel_max, n
        if b.size > 64:
            rel_norm = max(0.0, float(np.linalg.norm(a - b)) - 1e-7) / float(np.linalg.norm(b) + 1e-30)
            if rel_norm > worst["grad_norm"]:
                worst["grad_norm"], where["grad_norm"] = rel_norm, n
    for n, b1 in one["buffers"].items():
        assert torch.equal(r0["buffers"][n], r1["buffers"][n]), n
        a, b = r0["buffers"][n].numpy(), b1.numpy()
        rel = float((np.maximum(np.abs(a - b) - 1e-6, 0.0) / np.maximum(np.abs(b), 1e-30)).max()) if b.size else 0.0
        if rel > worst["buffer"]:
            worst["buffer"], where["buffer"] = rel, n
    return {"worst": worst, "where": where}


@pytest.mark.timeout(900)
@pytest.mark.parametrize("graph,dtype", [("auto", "fp32"), ("0", "fp32"), ("auto", "fp16")],
                         ids=["split_graphs", "eager", "split_graphs_fp16"])
def test_bench_two_ranks_on_one_gpu_emits_a_valid_line(graph, dtype):
    """The N > 1 leg of bench.py itself (the command the driver launches for SCALE), as two ranks sharing
    cuda:0 over gloo: launch mode, aggregate accounting, the parity deltas and the dist record."""
    import json

    # the literal command form the driver runs for SCALE (`python bench.py --gpus N ...`, no launcher, no WORLD_SIZE):
    # bench.py spawns its own ranks; the two test-only variables put both on cuda:0 over gloo (RCCL refuses that)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--resnet", "18",
           "--pairs", "8", "--size", "64", "--graph", graph, "--dtype", dtype]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PECLR_DIST_BACKEND="gloo", PECLR_SHARE_DEVICE="1")
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    details = [ln for ln in p.stdout.splitlines() if ln.startswith("BENCH_DETAILS ")]
    assert len(lines) == 1 and len(details) == 1 and p.stdout.rstrip().splitlines()[-1] == lines[0]
    assert len(lines[0]) < 4096
    kernels = json.loads(details[0][len("BENCH_DETAILS "):])["kernels"]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak" and r["unit"] == "images/sec"
    assert r["dtype"] == dtype               # fp16: the scaled gradients are all-reduced, every rank takes the same skip decision
    amp = r["config"].get("amp")
    assert (amp is None) == (dtype != "fp16")
    if amp is not None:
        assert 0 <= amp["optimizer_steps_taken_in_timed_region"] <= 3 and 0 < amp["loss_scale"] <= 65536.0
    assert r["config"]["global_batch"] == 2 * 2 * 8 and r["config"]["parallelism"] == "dp2"
    assert r["value"] == pytest.approx(2 * 2 * 8 * 3 / (r["ms_per_step"] * 3e-3), rel=1e-3)   # whole-job aggregate
    assert np.isfinite(r["loss"]) and 0 < r["loss"] < 10
    assert ("four hipGraph replays" in r["config"]["launch"]) == (graph == "auto")
    assert r["loss_delta_vs_oracle"] <= 1e-4 and r["sim_max_abs_delta"] <= 1e-4
    d = r["dist"]
    assert d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["grad_buckets"] >= 1
    assert "cpu_baseline" not in r           # rank 0 at N = 1 only
    assert r["roofline"]["kernel"] in kernels
    assert "self-spawned" in d["launcher"] and d["devices"] == [0, 0]
