"""16-bit 3x3 weight gradients: stride 1 at ResNet-50's four shapes, stride 2 at its three (first block of layers 2 - 4), vs MIOpen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi as capi
DEV = "cuda:0"
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
g = torch.Generator(device=DEV).manual_seed(0)
for hw, c, stride in ((56, 64, 1), (28, 128, 1), (14, 256, 1), (7, 512, 1), (28, 128, 2), (14, 256, 2), (7, 512, 2)):
    x = torch.randn(256, c, hw * stride, hw * stride, device=DEV, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(256, c, hw, hw, device=DEV, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    wb = torch.zeros(c, c, 3, 3, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t_h = timeit(lambda: capi.wgrad_h(gy, x, 9, stride))
    t_mi = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, wb, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    fl = 18 * c * c * 256 * hw * hw
    print(f"3x3 wgrad stride {stride} out {hw}x{hw} C={c}: in-tree {t_h:7.1f} us ({fl / t_h / 1e6:6.1f} TF, incl. slab reduce) | MIOpen bf16 {t_mi:7.1f} us", flush=True)
