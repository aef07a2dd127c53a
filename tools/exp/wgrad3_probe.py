"""3x3 weight gradients at ResNet-50's shapes (2 x 128 views @224): fp32 ring kernel vs the nine-splits kernel; 16-bit ring kernel vs MIOpen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi as capi
DEV = "cuda:0"

def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

g = torch.Generator(device=DEV).manual_seed(0)
for hw, c in ((56, 64), (28, 128), (14, 256), (7, 512)):
    x = torch.randn(256, c, hw, hw, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(256, c, hw, hw, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    gy2, x2 = gy.permute(0, 2, 3, 1).reshape(-1, c), x.permute(0, 2, 3, 1).reshape(-1, c)
    t_new = timeit(lambda: capi.wgrad3_x6r(gy, x))
    t_old = timeit(lambda: capi.gemm_x6t(gy2, x2, taps=9, hw=(hw, hw)))
    fl = 18 * c * c * 256 * hw * hw
    xb, gb = x.bfloat16(), gy.bfloat16()
    wb = torch.zeros(c, c, 3, 3, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t_h = timeit(lambda: capi.wgrad_h(gb, xb, 9, 1))
    t_mi = timeit(lambda: torch.ops.aten.convolution_backward(gb, xb, wb, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    print(f"3x3 wgrad {hw}x{hw} C={c}: fp32 ring {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF) | nine-splits {t_old:7.1f} us ({fl / t_old / 1e6:6.1f} TF) || bf16 ring {t_h:7.1f} us ({fl / t_h / 1e6:6.1f} TF) | MIOpen bf16 {t_mi:7.1f} us", flush=True)
