"""Per-launch time of the 16-bit 3x3 convolutions (forward with statistics, input gradient) at ResNet-50's four layer shapes
(2 x 128 views @224): us, TFLOP/s.   python tools/exp/conv3_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from peclr_amd import _capi as capi

DEV = "cuda:0"
dt = torch.bfloat16


def timeit(fn, reps=20):
    for _ in range(3):
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
    x = torch.randn(256, c, hw, hw, device=DEV, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(c, c, 3, 3, device=DEV, generator=g) * 0.03
    wf = w.permute(0, 2, 3, 1).reshape(c, 9 * c).contiguous()                 # [Cout][tap][Cin]
    wd = wf.reshape(c * 9, c)                                                 # [Cout * 9][Cin], packed with transposed = 9
    pk = capi.HPlanes([(wf, False), (wd, 9)], dt).pack()
    shift = torch.zeros(c, device=DEV)
    fl = 2 * 256 * hw * hw * 9 * c * c
    t = timeit(lambda: capi.conv_h(x, pk.planes[0], c, stat_shift=shift))
    t2 = timeit(lambda: capi.conv_h(x, pk.planes[1], c, flip=True))
    print(f"3x3 {hw}x{hw} x {c}: forward + stats {t:7.1f} us {fl / t / 1e6:6.1f} TFLOP/s | input gradient {t2:7.1f} us {fl / t2 / 1e6:6.1f} TFLOP/s")
