"""peclr_bn2d_apply / _bwd_apply (bf16) per layer shape of ResNet-50 at 2 x 128 views: us and TB/s; a device-to-device copy of the
same bytes beside them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi as capi
DEV = "cuda:0"; dt = torch.bfloat16
L = capi.lib()
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
io = capi._IO[dt][0]
for r, c in ((802816, 64), (802816, 256), (200704, 128), (200704, 512), (50176, 256), (50176, 1024), (12544, 512), (12544, 2048)):
    x = torch.randn(r, c, device=DEV).to(dt); y = torch.empty_like(x); res = torch.randn(r, c, device=DEV).to(dt)
    dy = torch.randn(r, c, device=DEV).to(dt); dx = torch.empty_like(x)
    ss = torch.randn(2, c, device=DEV); save = torch.rand(2, c, device=DEV) + 0.5; coef = torch.randn(2, c, device=DEV) * 0.01
    mask = torch.empty(r, c // 32, device=DEV, dtype=torch.int32)
    s = capi._stream()
    t1 = timeit(lambda: L.peclr_bn2d_apply(x.data_ptr(), None, io, r, c, ss.data_ptr(), 1, y.data_ptr(), None, s))
    t2 = timeit(lambda: L.peclr_bn2d_apply(x.data_ptr(), res.data_ptr(), io, r, c, ss.data_ptr(), 1, y.data_ptr(), mask.data_ptr(), s))
    t3 = timeit(lambda: L.peclr_bn2d_bwd_apply(dy.data_ptr(), x.data_ptr(), None, None, io, r, c, 1, save[0].data_ptr(), save[1].data_ptr(), ss.data_ptr(), coef.data_ptr(), dx.data_ptr(), None, s))
    t4 = timeit(lambda: L.peclr_bn2d_bwd_apply(dy.data_ptr(), x.data_ptr(), None, mask.data_ptr(), io, r, c, 1, save[0].data_ptr(), save[1].data_ptr(), ss.data_ptr(), coef.data_ptr(), dx.data_ptr(), res.data_ptr(), s))
    t0 = timeit(lambda: y.copy_(x))
    b = 2 * r * c
    print(f"[{r:6d}, {c:4d}] copy {t0:6.1f} us {2 * b / t0 / 1e6:5.2f} TB/s | apply {t1:6.1f} us {2 * b / t1 / 1e6:5.2f} | apply + residual + mask {t2:6.1f} us {(3 * b + r * c / 8) / t2 / 1e6:5.2f} | "
          f"bwd apply (recompute) {t3:6.1f} us {3 * b / t3 / 1e6:5.2f} | bwd apply (mask, dres) {t4:6.1f} us {(4 * b + r * c / 8) / t4 / 1e6:5.2f}")
