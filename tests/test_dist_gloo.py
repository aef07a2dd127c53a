"""N > 1 data-parallel path on CPU: world_size 2 over gloo (127.0.0.1), kernels replaced by the
oracle-backed double.  Checks SURVEY.md section 8e's contract: p ranks x N_local pairs give the same
loss and the same summed parameter gradients as one process on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import _capi_double

N_LOCAL, DIN, HID = int(os.environ.get("PECLR_TEST_N_LOCAL", "4")), 24, 32


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Patch:
    """monkeypatch stand-in for subprocesses."""

    def setattr(self, obj, name, value):
        setattr(obj, name, value)


class Encoder(torch.nn.Module):
    """Sample-wise encoder (no batch statistics) so 1-rank vs 2-rank equality is exact."""

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 3, 3, padding=1).to(memory_format=torch.channels_last)  # NHWC weight
        self.lin = torch.nn.Linear(3 * 4 * 4, DIN)
        self.final_layer = torch.nn.Linear(DIN, 4)  # never used: like encoder.final_layer

    def forward(self, x):
        return torch.tanh(self.lin(self.conv(x).flatten(1)))


def make_model(bn_eval):
    from peclr_amd import Config, Hybrid2Model

    torch.manual_seed(11)
    cfg = Config(projection_head_input_dim=DIN, projection_head_hidden_dim=HID, output_dim=128,
                 augmentation=["crop", "rotate"], batch_size=N_LOCAL, num_samples=64, num_of_mini_batch=1, lr=1e-3,
                 opt_weight_decay=1e-6, warmup_epochs=1, optimizer="LARS")
    model = Hybrid2Model(cfg)
    model.encoder = Encoder()
    model.train()
    if bn_eval:  # batch-independent head: isolates the collective logic from local-BN semantics
        model.projection_head[1].eval()
    return model


def make_batch(world):
    g = torch.Generator().manual_seed(3)
    n = world * N_LOCAL
    return {"transformed_image1": torch.randn(n, 3, 4, 4, generator=g),
            "transformed_image2": torch.randn(n, 3, 4, 4, generator=g),
            "jitter_x_1": torch.randint(-14, 1, (n,), generator=g), "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
            "jitter_y_1": torch.randint(-14, 1, (n,), generator=g), "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
            "angle_1": torch.randint(-45, 46, (n,), generator=g).double(),
            "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}


def shard(batch, rank):
    sl = slice(rank * N_LOCAL, (rank + 1) * N_LOCAL)
    return {k: v[sl].contiguous() for k, v in batch.items()}


def worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    _capi_double.install(_Patch())
    from peclr_amd import Trainer
    from peclr_amd import dist as pdist

    pdist.init_from_env(backend="gloo")
    assert pdist.world_size() == world and pdist.rank() == rank
    model = make_model(bn_eval=True)
    if rank == 1:  # ranks start different; attach() must broadcast rank 0's state
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    tr = Trainer(max_epochs=1, bucket_bytes=2048).attach(model)  # tiny buckets -> several all-reduces
    assert len(tr.reducer.buckets) > 1
    tr.zero_grad()
    batch = shard(make_batch(world), rank)
    tr.reducer.prepare(tr._unused)
    out = model.training_step(batch, 0)
    out["loss"].backward()
    tr.reducer.finish()
    for p in model.parameters():  # grads live in the flat buckets with the PARAMETER's strides
        assert p.grad.stride() == p.stride() and p.grad.untyped_storage().data_ptr() in {
            b.flat.untyped_storage().data_ptr() for b in tr.reducer.buckets}
    assert model.encoder.conv.weight.is_contiguous(memory_format=torch.channels_last)
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    torch.save({"loss": out["loss"].detach(), "grads": grads,
                "stats": {k: v for k, v in out.items() if k != "loss"}}, os.path.join(out_dir, f"r{rank}.pt"))
    # a rank whose batch has a different number of pairs is an error on EVERY rank, not a hang
    ragged = {k: v[: N_LOCAL - rank] for k, v in batch.items()}
    with pytest.raises(RuntimeError, match="differs across data-parallel ranks"):
        tr._check_uniform_batch(ragged)
    # ... also when the counts were equal before and only ONE rank's changes (rank 0 keeps its usual batch): the check
    # is a collective every rank enters on every micro-batch, not only the ranks whose own count moved
    tr._check_uniform_batch(batch)
    with pytest.raises(RuntimeError, match="differs across data-parallel ranks"):
        tr._check_uniform_batch(ragged if rank == 1 else batch)
    tr._check_uniform_batch(batch)
    # a full optimiser step keeps the replicas identical
    tr.optimizer.step()
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], t) for t in gathered[1:])
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,n_local", [(2, 4), (8, 2)])
def test_ranks_equal_one_rank(tmp_path, monkeypatch, world, n_local):
    """2 ranks x 4 pairs and 8 ranks x 2 pairs (the node size of BASELINE.json's configs 3 and 5: partner map, packed
    lse / statistics gather and bucket plan at the real world size) against one process on the concatenated batch."""
    global N_LOCAL
    monkeypatch.setenv("PECLR_TEST_N_LOCAL", str(n_local))     # the spawned workers re-import this module
    N_LOCAL = n_local
    mp.spawn(worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    for r in range(1, world):
        rk = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert torch.equal(r0["loss"], rk["loss"])  # every rank holds the global loss
        for n in r0["grads"]:
            assert torch.equal(r0["grads"][n], rk["grads"][n]), (r, n)  # SUM-reduced: identical everywhere

    # single process on the concatenated batch, rows reordered to the reference layout [all v1; all v2]
    _capi_double.install(monkeypatch)
    model = make_model(bn_eval=True)
    out = model.training_step(make_batch(world), 0)
    out["loss"].backward()
    assert abs(float(out["loss"].detach()) - float(r0["loss"])) < 2e-6
    # the 16 projection statistics are those of the GLOBAL batch on every rank (they ride in the packed gather)
    for k, v in r0["stats"].items():
        assert torch.equal(v, r1["stats"][k]), k
        assert abs(float(v) - float(out[k])) < 1e-6, k
    for n, p in model.named_parameters():
        if p.grad is None:
            assert float(r0["grads"][n].abs().max()) == 0.0, n  # final_layer: never gets a gradient
            continue
        ref = p.grad.numpy()
        np.testing.assert_allclose(r0["grads"][n].numpy(), ref, rtol=0, atol=1e-5 * max(1.0, np.abs(ref).max()),
                                   err_msg=n)


def test_single_process_helpers_are_noops():
    from peclr_amd import dist as pdist

    t = torch.arange(6.0).view(3, 2)
    assert pdist.world_size() == 1 and pdist.rank() == 0
    assert pdist.all_gather_cat(t) is t
    lin = torch.nn.Linear(2, 2)
    red = pdist.GradReducer(lin.parameters())
    red.prepare()
    lin(t).sum().backward()
    red.finish()
    assert lin.weight.grad.data_ptr() == red.buckets[0].views[-1].data_ptr() or \
        lin.weight.grad.data_ptr() == red.buckets[0].views[0].data_ptr()
    red.zero_grad()
    assert float(lin.weight.grad.abs().sum()) == 0.0


def test_grad_buckets_never_mix_backward_stages():
    """GradReducer(stage_of=...): a bucket closes where the backward stage changes (head + layer4 | the rest), so the
    first stage's buckets can be all-reduced while the second stage's backward is still running."""
    from peclr_amd import Hybrid2Model, Trainer, hybrid2_config
    from peclr_amd import dist as pdist

    import warnings

    warnings.simplefilter("ignore")
    cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, batch_size=2, num_samples=8, pretrained=False)
    model = Hybrid2Model(cfg)
    stage_of = Trainer._backward_stage_of(model)
    red = pdist.GradReducer(model.parameters(), None, bucket_bytes=4 << 20, stage_of=stage_of)
    stages = [b.stage for b in red.buckets]
    assert stages == sorted(stages) and set(stages) == {0, 1, 2} and len(red.buckets) > 3
    layer3 = {id(p) for p in model.encoder.features[6].parameters()}
    early = {id(p) for p in model.encoder.features[:6].parameters()}
    for b in red.buckets:
        for p in b.params:
            assert b.stage == (1 if id(p) in layer3 else 2 if id(p) in early else 0)
    assert sum(len(b.params) for b in red.buckets) == sum(1 for p in model.parameters() if p.requires_grad)
    assert red.launch(red.buckets) == []          # single process: nothing to reduce
    # an encoder that is not the in-tree wrapper has no seam: one stage
    model.encoder = torch.nn.Flatten()
    assert Trainer._backward_stage_of(model) is None
