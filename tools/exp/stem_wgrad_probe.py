"""Time peclr_stem_wgrad / peclr_stem_conv7x7_s2 at C2's shape (256 x 224 x 224) in isolation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peclr_amd import _capi
dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "fp32"]
x = torch.randn(256, 3, 224, 224, device="cuda").contiguous(memory_format=torch.channels_last)
gy = torch.randn(256, 64, 112, 112, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
w = torch.randn(64, 3, 7, 7, device="cuda") * 0.05
pl = _capi.StemPlanes(w, dt).pack()
def timeit(fn, reps=6):
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts[1:])[len(ts[1:]) // 2]
print(os.environ.get("PECLR_STEM_WGRAD_ABL", "0"), "wgrad %.1f us   fwd %.1f us" % (timeit(lambda: _capi.stem_wgrad(gy, x)), timeit(lambda: _capi.stem_conv(x, pl))))
