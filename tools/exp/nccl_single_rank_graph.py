# Experiment: do the split hipGraphs coexist with a live RCCL process group (watchdog thread, RCCL streams)?
# RCCL accepts a one-rank group on one GPU, which is all a single-GPU box can offer: collectives are issued
# before the capture, between the two graphs of every replay and after it, on the capture's side stream.
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter("ignore")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from peclr_amd import Hybrid2Model, Trainer, hybrid2_config
from peclr_amd.bn2d import enable_hip_batchnorm

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
print("backend", dist.get_backend(), flush=True)
n = 16
cfg = hybrid2_config(resnet_size="50", projection_head_input_dim=2048, augmentation=["crop", "rotate"], batch_size=n,
                     num_samples=64 * n, pretrained=False)
model = Hybrid2Model(cfg).cuda().train()
model.encoder = model.encoder.to(memory_format=torch.channels_last)
enable_hip_batchnorm(model.encoder)
tr = Trainer(max_epochs=10, grad_buckets=True, process_group=dist.group.WORLD).attach(model)
g = torch.Generator().manual_seed(1)
batch = {"transformed_image1": torch.randn(n, 3, 224, 224, generator=g), "transformed_image2": torch.randn(n, 3, 224, 224, generator=g),
         "jitter_x_1": torch.randint(-14, 1, (n,), generator=g), "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
         "jitter_y_1": torch.randint(-14, 1, (n,), generator=g), "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
         "angle_1": torch.randint(-45, 46, (n,), generator=g).double(), "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}
batch = {k: v.cuda() for k, v in batch.items()}
for k in ("transformed_image1", "transformed_image2"):
    batch[k] = batch[k].contiguous(memory_format=torch.channels_last)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    t = torch.ones(1 << 20, device="cuda")
    dist.all_reduce(t); out = torch.empty_like(t); dist.all_gather_into_tensor(out, t)
    tr.capture_split_graphs(batch, warmup=3)
    print("capture ok", flush=True)
    orig = model._contrast
    def contrast_with_collectives(z, nn_, rows):          # what world_size > 1 does between the graphs
        zz = torch.empty_like(z); dist.all_gather_into_tensor(zz, z.detach())
        loss = orig(z, nn_, rows)
        dist.all_reduce(t, async_op=True).wait()
        return loss
    model._contrast = contrast_with_collectives
    for i in range(5):
        o = tr.replay_split()
    for b in tr.reducer.buckets:
        dist.all_reduce(b.flat)
    torch.cuda.synchronize()
    print("replay ok", float(o["loss"]), flush=True)
dist.barrier(); dist.destroy_process_group()
print("done", flush=True)
