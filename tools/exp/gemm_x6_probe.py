"""peclr_gemm_x6_f32 (fp32 GEMM as six bf16 MFMA products) vs the v_mfma_f32 kernel: time and error against float64
at the backbone's GEMM shapes (2 x 128 views @224)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peclr_amd import _capi  # noqa: E402

# (M, N, K, addend): fork GEMM dX = dY W + dRes (K = Cmid, N = Cin); conv3 forward (K = Cmid, N = 4 Cmid);
# conv1 forward / conv3 dgrad (K = 4 Cmid, N = Cmid)
SHAPES = [(256 * 56 * 56, 256, 64, True), (256 * 28 * 28, 512, 128, True), (256 * 14 * 14, 1024, 256, True),
          (256 * 7 * 7, 2048, 512, True), (256 * 56 * 56, 256, 64, False), (256 * 14 * 14, 1024, 256, False),
          (256 * 56 * 56, 64, 256, False), (256 * 14 * 14, 256, 1024, False), (256 * 7 * 7, 512, 2048, False)]


def timeit(fn):
    junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(10):
        junk.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


for m, n, k, add in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    a = torch.randn(m, k, device="cuda", generator=g)
    bt = torch.randn(n, k, device="cuda", generator=g) * 0.05
    b = bt.t().contiguous()
    d = torch.randn(m, n, device="cuda", generator=g) if add else None
    x6 = _capi.gemm_x6(a, bt, d)
    f32 = _capi.gemm_add(_capi.GEMM_NN, a, b, d) if add else _capi.gemm(_capi.GEMM_NN, a, b)
    rows = torch.randint(0, m, (512,), device="cuda", generator=g)
    ref = a[rows].double() @ bt.double().t() + (d[rows].double() if add else 0)
    scale = float(ref.abs().max())
    e6 = float((x6[rows].double() - ref).abs().max()) / scale
    e32 = float((f32[rows].double() - ref).abs().max()) / scale
    t6 = timeit(lambda: _capi.gemm_x6(a, bt, d))
    t32 = timeit((lambda: _capi.gemm_add(_capi.GEMM_NN, a, b, d)) if add else (lambda: _capi.gemm(_capi.GEMM_NN, a, b)))
    tmi = timeit(lambda: torch.nn.functional.linear(a, bt))        # hipBLASLt fp32, no addend
    fl = 2 * m * n * k
    print(f"M={m:7d} N={n:5d} K={k:5d} add={int(add)}  x6 {t6:7.1f} us ({fl / t6 / 1e6:6.1f} TF)  mfma_f32 {t32:7.1f} us ({fl / t32 / 1e6:6.1f} TF)"
          f"  hipBLASLt {tmi:7.1f} us   max err / scale: x6 {e6:.2e}  f32 {e32:.2e}")
