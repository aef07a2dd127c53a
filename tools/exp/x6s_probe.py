"""The entry-gradient GEMM at layer1 / layer2's shapes: tiled (gemm_x6p) vs streaming (gemm_x6s) kernel, time and achieved HBM rate."""
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
for m, k, n in [(802816, 64, 256), (802816, 128, 256), (200704, 128, 512)]:
    g = torch.Generator(device="cuda").manual_seed(m + k + n)
    a = torch.randn(m, k, device="cuda", generator=g)
    bt = torch.randn(n, k, device="cuda", generator=g) * 0.05
    d = torch.randn(m, n, device="cuda", generator=g)
    xb = torch.randn(m, n, device="cuda", generator=g)
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (m, n // 32), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
    save = torch.stack([xb.mean(0), 1.0 / (xb.var(0, unbiased=False) + 1e-5).sqrt()]).contiguous()
    ss = torch.stack([save[1], -save[0] * save[1]]).contiguous()
    pk = _capi.X6Planes([(bt, False)]).pack().planes[0]
    link = (xb.view(1, m, 1, n).permute(0, 3, 1, 2), save, ss, mask, True)
    by = {"plain": 4 * m * (k + n), "+addend": 4 * m * (k + 2 * n), "+masked +bn bwd": 4 * m * (k + 3 * n) + m * n // 4}
    print(f"M={m} K={k} N={n}")
    for name, kw in (("plain", {}), ("+addend", {"addend": d}), ("+masked +bn bwd", {"addend": d, "addend_mask": mask, "bn_bwd": link})):
        for fam, fn in (("x6p", _capi.gemm_x6p), ("x6s", _capi.gemm_x6s)):
            t = timeit(lambda: fn(a, pk, n, **kw))
            print(f"    {name:18s} {fam}  {t:7.1f} us   {by[name] / 1e6:7.0f} MB  {by[name] / t / 1e6:5.2f} TB/s", flush=True)
