"""Run-to-run noise of the accumulated gradient (two identical eager steps) with and without MIOpen's deterministic attribute."""
import copy, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, ".miopen", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(ROOT, ".miopen", "cache"))
import torch
warnings.simplefilter("ignore")
from peclr_amd import Hybrid2Model, Trainer, hybrid2_config
from peclr_amd.bn2d import enable_hip_batchnorm
from bench import synthetic_batch

def build(resnet, pairs):
    cfg = hybrid2_config(resnet_size=resnet, projection_head_input_dim=2048, augmentation=["crop", "rotate"], batch_size=pairs, pretrained=False)
    torch.manual_seed(5)
    m = Hybrid2Model(cfg).cuda().train()
    m.encoder = m.encoder.to(memory_format=torch.channels_last)
    enable_hip_batchnorm(m.encoder)
    return m

resnet = sys.argv[1] if len(sys.argv) > 1 else "50"
base = build(resnet, 128)
batch = synthetic_batch(128, 224, 5, torch.device("cuda"), channels_last=True)
def grads():
    model = copy.deepcopy(base)
    model.zero_grad(set_to_none=True)
    out = model.training_step(batch, 0)
    out["loss"].backward()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in model.parameters() if p.grad is not None]
def dev(a, b):
    num = sum(float((x - y).double().pow(2).sum()) for x, y in zip(a, b)); den = sum(float(x.double().pow(2).sum()) for x in a)
    return (num / den) ** 0.5
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    t0 = time.time(); g0 = grads(); t1 = time.time(); g1 = grads(); g2 = grads(); t2 = time.time()
    print(f"resnet{resnet} deterministic={det}: run-to-run {dev(g0, g1):.3e} {dev(g1, g2):.3e}  first {t1 - t0:.1f}s next {(t2 - t1) / 2:.2f}s", flush=True)
