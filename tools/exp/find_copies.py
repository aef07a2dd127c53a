"""Which Python call sites issue device-to-device copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer) in one eager bf16 step?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "--dtype", os.environ.get("DT", "bf16"), "--no-cpu-baseline"]
import torch
import bench
from peclr_amd import Trainer
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args, dev, args.pairs)
model.encoder = model.encoder.to(memory_format=torch.channels_last)
from peclr_amd.bn2d import enable_hip_batchnorm
enable_hip_batchnorm(model.encoder)
trainer = Trainer(max_epochs=100, accumulate_grad_batches=1, precision=args.dtype).attach(model)
trainer.zero_grad()
batch = bench.synthetic_batch(args.pairs, args.size, 5, dev, channels_last=True)
for i in range(3):
    trainer.training_micro_step(batch, i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    trainer.training_micro_step(batch, 3)
    torch.cuda.synchronize()
cnt = collections.Counter()
shapes = collections.defaultdict(list)
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy") and ev.device_type == torch.autograd.DeviceType.CPU:
        st = [f for f in (ev.stack or []) if "peclr_amd" in f or "bench" in f or "torch/autograd" in f][:3]
        key = (ev.name, " <- ".join(s.split("/")[-1] for s in st) or "(no python frame: autograd engine)")
        cnt[key] += 1
        if len(shapes[key]) < 3:
            shapes[key].append(str(ev.input_shapes)[:80])
for (name, where), n in cnt.most_common(25):
    print(f"{n:4d}  {name:16s} {where[:170]}   e.g. {shapes[(name, where)][:2]}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8, max_name_column_width=60)[:3000])
