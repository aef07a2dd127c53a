"""layer1's 1x1 convolutions (64 <-> 256 channels, 802 816 rows): peclr_gemm_x6p_f32 (64-column tiles where N = 64) against MIOpen."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".miopen", "db"))
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
n, hw = 256, 56
for cin, cout in ((64, 256), (256, 64), (64, 64)):
    x = torch.randn(n, cin, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, cout, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    w2 = w.reshape(cout, cin)
    pk = _capi.X6Planes([(w2, False), (w2, True)]).pack()
    r = n * hw * hw
    x2, gy2 = x.permute(0, 2, 3, 1).reshape(r, cin), gy.permute(0, 2, 3, 1).reshape(r, cout)
    shift = torch.zeros(cout, device="cuda")
    y = _capi.gemm_x6p(x2, pk.planes[0], cout)
    ref = F.conv2d(x, w).permute(0, 2, 3, 1).reshape(r, cout)
    print(f"{cin}->{cout}: fwd max diff vs MIOpen {float((y - ref).abs().max()):.2e}", end="  ")
    for tr in (128, 256):
        print(f"fwd x6p/{tr} {timeit(lambda: _capi.gemm_x6p(x2, pk.planes[0], cout, tile_rows=tr)):6.1f}", end=" ")
    print(f"+stats {timeit(lambda: _capi.gemm_x6p(x2, pk.planes[0], cout, stat_shift=shift)):6.1f}  MIOpen {timeit(lambda: F.conv2d(x, w)):6.1f} us", end=" | ")
    for tr in (128, 256):
        print(f"dgrad x6p/{tr} {timeit(lambda: _capi.gemm_x6p(gy2, pk.planes[1], cin, tile_rows=tr)):6.1f}", end=" ")
    print(f"MIOpen {timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])):6.1f} us", end=" | ")
    print(f"wgrad x6t {timeit(lambda: _capi.gemm_x6t(gy2, x2)):6.1f} x6_tn {timeit(lambda: _capi.gemm_x6_tn(gy2, x2)):6.1f} MIOpen {timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])):6.1f} us", flush=True)
