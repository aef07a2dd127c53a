"""peclr_bn2d_finalize_f32 / _bwd_finalize_f32 per (partial rows, channels) of ResNet-50's 16-bit step (128-row GEMM tiles; ring 3x3: 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi as capi
DEV = "cuda:0"
L = capi.lib()
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
tot = 0.0
for rows, c, count in ((1605632, 64, 3), (1605632, 256, 4), (802816, 64, 3), (802816, 256, 4), (200704, 128, 4), (200704, 512, 5), (50176, 256, 6), (50176, 1024, 7), (12544, 512, 3), (12544, 2048, 4),
                       (802816 * 57 * 57 // (56 * 56) // 2, 64, 3), (200704 * 29 * 29 // (28 * 28) // 2, 128, 3), (50176 * 15 * 15 // (14 * 14) // 2, 256, 5), (12544 * 64 // 49 // 2, 512, 2)):
    ns = (rows + 127) // 128
    part = torch.randn(2 * ns + 1, c, device=DEV)
    gamma, beta = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    nbt = torch.zeros((), device=DEV, dtype=torch.int64)
    save, ss = torch.empty(2, c, device=DEV), torch.empty(2, c, device=DEV)
    dp, coef = torch.empty(2, c, device=DEV), torch.empty(2, c, device=DEV)
    s = capi._stream()
    t1 = timeit(lambda: L.peclr_bn2d_finalize_f32(part.data_ptr(), ns, rows, c, 1, 1e-5, 0.1, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), save[0].data_ptr(), save[1].data_ptr(), ss.data_ptr(), s))
    t2 = timeit(lambda: L.peclr_bn2d_bwd_finalize_f32(part.data_ptr(), ns, rows, c, 1, ss.data_ptr(), dp[0].data_ptr(), dp[1].data_ptr(), coef.data_ptr(), s))
    tot += count * (t1 + t2)
    print(f"partial rows {ns:5d} x {c:4d} channels ({2 * ns * c * 4 / 1e6:6.2f} MB): finalize {t1:6.1f} us, backward finalize {t2:6.1f} us   (x {count} layers)")
print(f"sum over the step's layers (eager launches back to back): {tot / 1e3:.2f} ms")
