"""The entry-gradient GEMM (peclr_gemm_x6p_f32 and its epilogue variants) at layer1 / layer2's HBM-bound shapes: time and
achieved HBM rate of the plain product, + dense addend, + masked addend, + BatchNorm backward reduction."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peclr_amd import _capi
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
def timeit(fn, reps=8):
    ts = []
    for _ in range(reps):
        junk.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts = sorted(ts[2:]); return ts[len(ts) // 2]
for m, k, n in [(802816, 64, 256), (200704, 128, 512), (50176, 256, 1024)]:
    g = torch.Generator(device="cuda").manual_seed(m + k + n)
    a = torch.randn(m, k, device="cuda", generator=g)
    bt = torch.randn(n, k, device="cuda", generator=g) * 0.05
    d = torch.randn(m, n, device="cuda", generator=g)
    xb = torch.randn(m, n, device="cuda", generator=g)
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (m, n // 32), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
    save = torch.stack([xb.mean(0), 1.0 / (xb.var(0, unbiased=False) + 1e-5).sqrt()]).contiguous()
    ss = torch.stack([save[1], -save[0] * save[1]]).contiguous()
    pk = _capi.X6Planes([(bt, False)]).pack()
    link = (xb.view(1, 1, m, n).permute(0, 3, 1, 2), save, ss, mask, True)      # NHWC view [1, n, 1, m]: m rows of n channels
    runs = {"plain": (lambda: _capi.gemm_x6p(a, pk.planes[0], n), 4 * m * (k + n)),
            "+addend": (lambda: _capi.gemm_x6p(a, pk.planes[0], n, d), 4 * m * (k + 2 * n)),
            "+masked addend": (lambda: _capi.gemm_x6p(a, pk.planes[0], n, d, addend_mask=mask), 4 * m * (k + 2 * n) + m * n // 8),
            "+addend +bn bwd": (lambda: _capi.gemm_x6p(a, pk.planes[0], n, d, bn_bwd=link), 4 * m * (k + 3 * n) + m * n // 8),
            "+masked +bn bwd": (lambda: _capi.gemm_x6p(a, pk.planes[0], n, d, addend_mask=mask, bn_bwd=link), 4 * m * (k + 3 * n) + m * n // 4)}
    print(f"M={m} K={k} N={n}  (2MNK = {2 * m * n * k / 1e9:.1f} GF = {2 * m * n * k / 416.7e6:.0f} us at the MFMA roof)")
    for name, (fn, by) in runs.items():
        t = timeit(fn)
        print(f"    {name:18s} {t:7.1f} us   {by / 1e6:7.0f} MB  {by / t / 1e6:5.2f} TB/s", flush=True)
