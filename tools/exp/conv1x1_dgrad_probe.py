# Experiment: 1x1-conv input gradient of the bottleneck's first conv -- MIOpen dgrad + the autograd add of the
# residual gradient vs. ONE fp32 GEMM (dX = dY W) of libpeclr_hip at the four ResNet-50 stage shapes.
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), ".miopen", "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), ".miopen", "cache"))
import torch
from peclr_amd import _capi

def timed(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters

for (n, hw, cmid, cin) in ((256, 56, 64, 256), (256, 28, 128, 512), (256, 14, 256, 1024), (256, 7, 512, 2048)):
    x = torch.randn(n, cin, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cmid, cin, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, cmid, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    dres = torch.randn_like(x)
    f = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    t_dgrad = timed(f)
    g = f()
    t_add = timed(lambda: g.add_(dres))
    r = n * hw * hw
    a, b = dy.permute(0, 2, 3, 1).reshape(r, cmid), w.reshape(cmid, cin)
    t_gemm = timed(lambda: _capi.gemm(_capi.GEMM_NN, a, b))
    ref = f().permute(0, 2, 3, 1).reshape(r, cin)
    err = float((_capi.gemm(_capi.GEMM_NN, a, b) - ref).abs().max())
    print(f"R={r} Cmid={cmid} Cin={cin}: miopen dgrad {t_dgrad:.0f} us + add {t_add:.0f} us = {t_dgrad + t_add:.0f} us | peclr gemm NN {t_gemm:.0f} us (max |d| {err:.1e})", flush=True)

print("with the residual gradient added in the epilogue (peclr_gemm_add_f32):")
for (n, hw, cmid, cin) in ((256, 56, 64, 256), (256, 28, 128, 512), (256, 14, 256, 1024), (256, 7, 512, 2048)):
    r = n * hw * hw
    a, b, d = torch.randn(r, cmid, device="cuda"), torch.randn(cmid, cin, device="cuda"), torch.randn(r, cin, device="cuda")
    t = timed(lambda: _capi.gemm_add(_capi.GEMM_NN, a, b, d))
    fl, by = 2 * r * cmid * cin, 4 * (r * cmid + 2 * r * cin)
    print(f"R={r} Cmid={cmid} Cin={cin}: {t:.0f} us  {fl / t / 1e6:.1f} TFLOP/s  {by / t / 1e3:.0f} GB/s", flush=True)
