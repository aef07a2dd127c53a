"""Child process of tests/test_dist_rccl_gpu.py: a LIVE one-rank RCCL process group on cuda:0 (RCCL refuses two ranks per
device, so one rank is all a single-GPU box can host) with `peclr_amd.dist.FORCE_COLLECTIVES` on, i.e. every collective of
the data-parallel path is really issued through RCCL -- all_gather_into_tensor of the embeddings, the packed
lse / statistics / loss gather, the asynchronous SUM all-reduce of the stage-pure gradient buckets between the hipGraph
replays of the split step -- although each is the identity on one rank.  Checks: the split-graph step equals the eager
step and the plain single-process step (no process group) to fp32 round-off, five replays in a row."""
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29541")
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, ".miopen", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(ROOT, ".miopen", "cache"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from peclr_amd import Hybrid2Model, Trainer, hybrid2_config  # noqa: E402
from peclr_amd import dist as pdist  # noqa: E402
from peclr_amd.bn2d import enable_hip_batchnorm  # noqa: E402


def build(n):
    torch.manual_seed(5)
    cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, augmentation=["crop", "rotate"], batch_size=n,
                         num_samples=64 * n, pretrained=False)
    model = Hybrid2Model(cfg).cuda().train()
    model.encoder = model.encoder.to(memory_format=torch.channels_last)
    enable_hip_batchnorm(model.encoder)
    return model


def make_batch(n):
    g = torch.Generator().manual_seed(1)
    b = {"transformed_image1": torch.randn(n, 3, 96, 96, generator=g), "transformed_image2": torch.randn(n, 3, 96, 96, generator=g),
         "jitter_x_1": torch.randint(-14, 1, (n,), generator=g), "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
         "jitter_y_1": torch.randint(-14, 1, (n,), generator=g), "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
         "angle_1": torch.randint(-45, 46, (n,), generator=g).double(), "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}
    b = {k: v.cuda() for k, v in b.items()}
    for k in ("transformed_image1", "transformed_image2"):
        b[k] = b[k].contiguous(memory_format=torch.channels_last)
    return b


def main():
    n, steps = 8, 5
    torch.cuda.set_device(0)
    batch = make_batch(n)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    losses = {}
    # reference: no process group at all, eager steps
    with torch.cuda.stream(side):
        model = build(n)
        tr = Trainer(max_epochs=10, grad_buckets=True).attach(model)
        tr.zero_grad()
        losses["plain"] = [float(tr.training_micro_step(batch, i)["loss"]) for i in range(steps + 3)]
    torch.cuda.synchronize()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pdist.FORCE_COLLECTIVES = True
    counts = {"all_gather_into_tensor": 0, "all_reduce": 0}
    real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce

    def ag(*a, **k):
        counts["all_gather_into_tensor"] += 1
        return real_ag(*a, **k)

    def ar(*a, **k):
        counts["all_reduce"] += 1
        return real_ar(*a, **k)

    dist.all_gather_into_tensor, dist.all_reduce = ag, ar
    with torch.cuda.stream(side):
        # eager steps through RCCL
        model = build(n)
        tr = Trainer(max_epochs=10, grad_buckets=True, process_group=dist.group.WORLD).attach(model)
        tr.zero_grad()
        losses["eager_rccl"] = [float(tr.training_micro_step(batch, i)["loss"]) for i in range(steps + 3)]
        eager_counts = dict(counts)
        # the split hipGraphs (forward | three backward stages) with the RCCL collectives between their replays
        model = build(n)
        tr = Trainer(max_epochs=10, grad_buckets=True, process_group=dist.group.WORLD).attach(model)
        tr.zero_grad()
        tr.capture_split_graphs(batch, warmup=3)
        before = dict(counts)
        losses["graph_rccl"] = [float(tr.replay_split()["loss"]) for _ in range(steps)]
        torch.cuda.synchronize()
        per_replay = {k: (counts[k] - before[k]) / steps for k in counts}
        n_buckets, n_graphs = len(tr.reducer.buckets), 1 + len(tr._graph_bs)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"losses": losses, "eager_counts": eager_counts, "per_replay": per_replay, "buckets": n_buckets,
                      "graphs": n_graphs, "backend": "nccl", "rccl": ".".join(str(v) for v in torch.cuda.nccl.version())}))


if __name__ == "__main__":
    main()
