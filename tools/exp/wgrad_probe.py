"""peclr_gemm_x6t_f32 (weight gradients, 256 x 256 tiles, double-buffered k-step 16) against peclr_gemm_x6_tn_f32 and MIOpen:
error against float64 and time, 1x1 and 3x3 shapes of ResNet-50 (2 x 128 views)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".miopen", "db"))
from peclr_amd import _capi  # noqa: E402
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
def timeit(fn, reps=8):
    ts = []
    for _ in range(reps):
        junk.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts = sorted(ts[2:]); return ts[len(ts) // 2]
small = len(sys.argv) > 1
R2, R3, R4 = 256 * 28 * 28, 256 * 14 * 14, 256 * 7 * 7
S1 = [(R2, 128, 512), (R2, 512, 128), (R3, 256, 1024), (R3, 1024, 256), (R4, 512, 2048), (R4, 2048, 512), (R2, 128, 256)]
S3 = [(256, 64, 56), (256, 128, 28), (256, 256, 14), (256, 512, 7)]
if small:
    S1 = [(4100, 128, 256), (8192 + 4, 132, 260), (3000, 512, 128)]; S3 = [(6, 128, 9), (4, 256, 7), (5, 64, 10)]
for k, m, n in S1:
    g = torch.Generator(device="cuda").manual_seed(k + m + n)
    a = torch.randn(k, m, device="cuda", generator=g); b = torch.randn(k, n, device="cuda", generator=g)
    new, old = _capi.gemm_x6t(a, b), _capi.gemm_x6_tn(a, b)
    ref = a.double().t() @ b.double(); bound = a.double().abs().t() @ b.double().abs()
    en, eo = float(((new.double() - ref).abs() / bound).max()), float(((old.double() - ref).abs() / bound).max())
    det = torch.equal(_capi.gemm_x6t(a, b), new)
    tn, to = timeit(lambda: _capi.gemm_x6t(a, b)), timeit(lambda: _capi.gemm_x6_tn(a, b))
    fl = 2 * k * m * n
    print(f"1x1 K={k} M={m} N={n}: err/bound x6t {en:.2e} x6_tn {eo:.2e} deterministic {det} | x6t {tn:7.1f} us ({fl / tn / 1e6:5.1f} TF)  x6_tn {to:7.1f} us ({fl / to / 1e6:5.1f} TF)", flush=True)
for nb, c, hw in S3:
    g = torch.Generator(device="cuda").manual_seed(c + hw)
    x = torch.randn(nb, c, hw, hw, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(nb, c, hw, hw, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(c, c, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    r = nb * hw * hw
    gy2, x2 = gy.permute(0, 2, 3, 1).reshape(r, c), x.permute(0, 2, 3, 1).reshape(r, c)
    run = lambda: _capi.gemm_x6t(gy2, x2, taps=9, hw=(hw, hw))
    mi = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    dw = run().view(c, 3, 3, c).permute(0, 3, 1, 2)
    dw_mi = mi()
    sub = slice(0, min(nb, 256))
    ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1] if r <= 60000 else None
    if ref is not None:
        sc = float(ref.abs().max())
        print(f"3x3 N={nb} C={c} {hw}x{hw}: err/scale x6t {float((dw.double() - ref).abs().max()) / sc:.2e} MIOpen {float((dw_mi.double() - ref).abs().max()) / sc:.2e}", end=" ")
    else:
        print(f"3x3 N={nb} C={c} {hw}x{hw}: x6t vs MIOpen {float((dw - dw_mi).abs().max()) / float(dw_mi.abs().max()):.2e}", end=" ")
    tn, tm = timeit(run), timeit(mi)
    fl = 18 * r * c * c
    print(f"| x6t {tn:7.1f} us ({fl / tn / 1e6:5.1f} TF)  MIOpen {tm:7.1f} us ({fl / tm / 1e6:5.1f} TF)  deterministic {torch.equal(run(), run())}", flush=True)
