"""3x3 forward: time per k-step at small channel counts (weights stay in L2 between launches) vs large (they stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi as capi
DEV = "cuda:0"; dt = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
g = torch.Generator(device=DEV).manual_seed(0)
for hw, c, nb in ((7, 128, 256), (7, 256, 256), (7, 512, 256), (7, 512, 64), (14, 128, 256), (14, 256, 256), (14, 512, 256), (28, 128, 256), (28, 64, 256)):
    x = torch.randn(nb, c, hw, hw, device=DEV, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(c, 9 * c, device=DEV, generator=g) * 0.03
    pk = capi.HPlanes([(w, False)], dt).pack()
    for tr in (1, 128):
        t = timeit(lambda: capi.conv_h(x, pk.planes[0], c, tile_rows=tr))
        steps = 9 * c // 32
        rows = nb * (hw + 1) * (hw + 1) if tr == 1 else nb * hw * hw
        wgs = -(-rows // (256 if tr == 1 else 128)) * -(-c // 128)
        print(f"{hw}x{hw} x {c} nb {nb} tile {tr:3d}: {t:7.1f} us  {steps} steps {t / steps * 1e3:6.0f} ns/step  {wgs} workgroups  weights {18 * c * c / 1e6:.2f} MB  {2 * nb * hw * hw * 9 * c * c / t / 1e6:6.1f} TFLOP/s")
