# Experiment: cost of cutting the step into its N>1 form (forward graph + one backward graph per stage) vs one graph, on ONE GPU (no collectives):
# the extra eager launches between / after the graphs and the gradient copy into the flat buckets.
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT))
warnings.simplefilter("ignore")
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, ".miopen", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(ROOT, ".miopen", "cache"))
import torch
import bench
from peclr_amd import Trainer
from peclr_amd.bn2d import enable_hip_batchnorm

class A: resnet = "50"; accum = 1
mode = sys.argv[1]
dev = torch.device("cuda:0")
model = bench.build_model(A, dev, 128)
model.encoder = model.encoder.to(memory_format=torch.channels_last)
enable_hip_batchnorm(model.encoder)
tr = Trainer(max_epochs=100, grad_buckets=(mode == "split") or None).attach(model)
batch = bench.synthetic_batch(128, 224, 5, dev, channels_last=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    if mode == "split":
        tr.capture_split_graphs(batch, warmup=3); rep = tr.replay_split
    else:
        tr.capture_step_graph(batch, warmup=3); rep = tr.replay_step
    for _ in range(3): rep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): rep()
    torch.cuda.synchronize()
print(mode, round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms/step", flush=True)
