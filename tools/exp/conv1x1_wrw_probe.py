# Experiment: weight gradient of every 1x1 stride-1 convolution shape of ResNet-50 (2x128 views @224):
# MIOpen wrw (atomic split-K + zero fill) vs libpeclr_hip's TN GEMM with deterministic split-K slabs + slab reduce.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, ".miopen", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(ROOT, ".miopen", "cache"))
import torch
from peclr_amd import _capi

def timed(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters

shapes = [(56, 64, 64, 1), (56, 256, 64, 2), (56, 64, 256, 3), (56, 256, 128, 1), (28, 512, 128, 3), (28, 128, 512, 4),
          (28, 512, 256, 1), (14, 1024, 256, 5), (14, 256, 1024, 6), (14, 1024, 512, 1), (7, 2048, 512, 2), (7, 512, 2048, 3)]
tot = {"miopen": 0, "peclr": 0}
for hw, cin, cout, count in shapes:
    n = 256
    x = torch.randn(n, cin, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, cout, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    r = n * hw * hw
    xm, dym = x.permute(0, 2, 3, 1).reshape(r, cin), dy.permute(0, 2, 3, 1).reshape(r, cout)
    f = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    mi = timed(f)
    best = None
    for sk in (16, 32, 64, 128, 256, 512):
        tiles = ((cout + 63) // 64) * ((cin + 63) // 64) * sk
        if tiles > 16384 or r // sk < 64:
            continue
        def g():
            slabs = _capi.gemm(_capi.GEMM_TN, dym, xm, split_k=sk)          # [sk, cout, cin]
            return _capi.slab_reduce(slabs) if slabs.dim() == 3 else slabs
        t = timed(g)
        if best is None or t < best[0]:
            best = (t, sk)
    ref = f().reshape(cout, cin)
    sk = best[1]
    slabs = _capi.gemm(_capi.GEMM_TN, dym, xm, split_k=sk)
    got = _capi.slab_reduce(slabs) if slabs.dim() == 3 else slabs
    err = float((got - ref).abs().max() / ref.abs().max())
    tot["miopen"] += mi * count; tot["peclr"] += best[0] * count
    print(f"{hw:3d}x{hw:<3d} {cin:5d}->{cout:<5d} x{count}: wrw miopen {mi:5.0f} us | peclr TN split-K {best[1]:3d}: {best[0]:5.0f} us (rel err {err:.1e})", flush=True)
print({k: round(v / 1e3, 2) for k, v in tot.items()}, "ms per step")
