"""peclr_gemm_x6p_f32 (weight planes packed once, LDS-DMA) vs peclr_gemm_x6_f32 (both operands split per workgroup):
bit-equality of the two, error against float64, time at the backbone's 1x1 GEMM shapes (2 x 128 views @224)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peclr_amd import _capi  # noqa: E402

R2, R3, R4 = 256 * 28 * 28, 256 * 14 * 14, 256 * 7 * 7
# (M, N, K, addend): conv1 forward / conv3 dgrad (K = 4 Cmid -> N = Cmid), conv3 forward (K = Cmid -> N = 4 Cmid),
# fork dgrad (K = Cmid -> N = 4 Cmid, + addend)
SHAPES = [(R2, 128, 512, False), (R2, 512, 128, False), (R2, 512, 128, True),
          (R3, 256, 1024, False), (R3, 1024, 256, False), (R3, 1024, 256, True),
          (R4, 512, 2048, False), (R4, 2048, 512, False), (R4, 2048, 512, True)]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    SHAPES = [(1000, 128, 64, True), (256, 256, 16, False), (4096, 384, 160, True)]

junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def timeit(fn, reps=12):
    ts = []
    for _ in range(reps):
        junk.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], ts[0]


for m, n, k, add in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    a = torch.randn(m, k, device="cuda", generator=g)
    bt = torch.randn(n, k, device="cuda", generator=g) * 0.05
    d = torch.randn(m, n, device="cuda", generator=g) if add else None
    pk = _capi.X6Planes([(bt, False), (bt.t().contiguous(), True)]).pack()
    old = _capi.gemm_x6(a, bt, d) if k % 4 == 0 else None
    outs = {tr: _capi.gemm_x6p(a, pk.planes[0], n, d, tile_rows=tr) for tr in (256, 128)}
    out_t = _capi.gemm_x6p(a, pk.planes[1], n, d, tile_rows=256)          # planes packed from the transposed storage
    torch.cuda.synchronize()
    rows = torch.randint(0, m, (512,), device="cuda", generator=g)
    ref = a[rows].double() @ bt.double().t() + (d[rows].double() if add else 0)
    scale = float(ref.abs().max())
    err = {tr: float((o[rows].double() - ref).abs().max()) / scale for tr, o in outs.items()}
    same = {tr: bool(torch.equal(o, old)) for tr, o in outs.items()} if old is not None else None
    line = (f"M={m:7d} N={n:5d} K={k:5d} add={int(add)}  err/scale {err[256]:.2e} {err[128]:.2e}  bit-equal to x6: {same} "
            f" transposed-pack equal: {bool(torch.equal(out_t, outs[256]))}")
    fl = 2 * m * n * k
    t_old = timeit(lambda: _capi.gemm_x6(a, bt, d))
    t = {tr: timeit(lambda tr=tr: _capi.gemm_x6p(a, pk.planes[0], n, d, tile_rows=tr)) for tr in (256, 128)}
    t_pack = timeit(pk.pack)
    print(line)
    print(f"    x6 {t_old[0]:7.1f} us ({fl / t_old[0] / 1e6:6.1f} TF)   x6p/256 {t[256][0]:7.1f} us ({fl / t[256][0] / 1e6:6.1f} TF, min "
          f"{t[256][1]:6.1f})   x6p/128 {t[128][0]:7.1f} us ({fl / t[128][0] / 1e6:6.1f} TF, min {t[128][1]:6.1f})   pack x2 {t_pack[0]:5.1f} us",
          flush=True)
