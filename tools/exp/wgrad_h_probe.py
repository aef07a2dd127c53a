"""Per-launch time of peclr_wgrad_h at ResNet-50's 1x1 shapes (2 x 128 views @224) next to MIOpen's 16-bit weight gradient."""
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
for hw, cmid in ((56, 64), (28, 128), (14, 256), (7, 512)):
    for cout, cin in ((cmid, 4 * cmid), (4 * cmid, cmid), (cmid, cmid) if cmid == 64 else (4 * cmid, 2 * cmid)):
        x = torch.randn(256, cin, hw, hw, device=DEV, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(256, cout, hw, hw, device=DEV, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 1, 1, device=DEV, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
        t = timeit(lambda: capi.wgrad_h(gy, x))
        tm = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
        by = 2 * 256 * hw * hw * (cin + cout)
        print(f"wgrad 1x1 {hw}x{hw} [{cout} <- {cin}]: in-tree {t:7.1f} us {by / t / 1e6:5.2f} TB/s | MIOpen {tm:7.1f} us  (slabs {capi.lib().peclr_wgrad_h_slabs(cout, cin, 256 * hw * hw)})")
