"""Weight gradients that round 3 moved last: layer1's 64-wide shapes and the stride-2 convolutions of layers 2-4
(peclr_gemm_x6t_f32 with the 64-wide tiles / stride = 2) against MIOpen's fp32 kernels, 2 x 128 views @224."""
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
NB = 256
# (cin, cout, hin, ksize, stride)
CASES = [(64, 64, 56, 1, 1), (64, 256, 56, 1, 1), (256, 64, 56, 1, 1), (64, 64, 56, 3, 1),
         (256, 512, 56, 1, 2), (512, 1024, 28, 1, 2), (1024, 2048, 14, 1, 2),
         (128, 128, 56, 3, 2), (256, 256, 28, 3, 2), (512, 512, 14, 3, 2)]
for cin, cout, hin, ks, st in CASES:
    ho = hin // st
    pad = ks // 2
    g = torch.Generator(device="cuda").manual_seed(cin + cout + hin)
    x = torch.randn(NB, cin, hin, hin, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(NB, cout, ho, ho, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(cout, cin, ks, ks, device="cuda").contiguous(memory_format=torch.channels_last)
    gy2, x2 = gy.permute(0, 2, 3, 1).reshape(NB * ho * ho, cout), x.permute(0, 2, 3, 1).reshape(NB * hin * hin, cin)
    run = lambda: _capi.gemm_x6t(gy2, x2, taps=ks * ks, hw=(ho, ho), stride=st)
    mi = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    dw = run().view(cout, ks, ks, cin).permute(0, 3, 1, 2)
    dw_mi = mi()
    diff = float((dw - dw_mi).abs().max()) / float(dw_mi.abs().max())
    tn, tm = timeit(run), timeit(mi)
    fl = 2 * NB * ho * ho * cin * cout * ks * ks
    by = 4 * (gy.numel() + (x.numel() if not (st == 2 and ks == 1) else x.numel() // 4))
    print(f"{ks}x{ks}/s{st} {cin:4d}->{cout:4d} @{hin}: x6t vs MIOpen {diff:.1e} | x6t {tn:7.1f} us ({fl / tn / 1e6:5.1f} TF, {by / tn / 1e6:4.2f} TB/s)"
          f"  MIOpen {tm:7.1f} us   slabs {_capi.lib().peclr_gemm_x6t_slabs(cout, cin, NB * ho * ho, ks * ks)}", flush=True)
