# Experiment: every 1x1 stride-1 convolution shape of ResNet-50 (2x128 views @224): MIOpen forward / input-gradient
# time vs the fp32 GEMM of libpeclr_hip on the same NHWC storage.
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
tot = {"mi_f": 0, "my_f": 0, "mi_d": 0, "my_d": 0}
for hw, cin, cout, count in shapes:
    n = 256
    x = torch.randn(n, cin, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, cout, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    r = n * hw * hw
    xm, wm, dym = x.permute(0, 2, 3, 1).reshape(r, cin), w.reshape(cout, cin), dy.permute(0, 2, 3, 1).reshape(r, cout)
    mi_f = timed(lambda: torch.nn.functional.conv2d(x, w))
    my_f = timed(lambda: _capi.gemm(_capi.GEMM_NT, xm, wm))
    mi_d = timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0])
    my_d = timed(lambda: _capi.gemm(_capi.GEMM_NN, dym, wm))
    for k, v in (("mi_f", mi_f), ("my_f", my_f), ("mi_d", mi_d), ("my_d", my_d)): tot[k] += v * count
    print(f"{hw:3d}x{hw:<3d} {cin:5d}->{cout:<5d} x{count}: fwd miopen {mi_f:5.0f} peclr {my_f:5.0f} | dgrad miopen {mi_d:5.0f} peclr {my_d:5.0f} us", flush=True)
print({k: round(v / 1e3, 2) for k, v in tot.items()}, "ms per step (weighted by the number of such layers)")
