"""peclr_conv3x3_x6p_f32 (3x3 convolution as an implicit GEMM, six bf16 MFMA products per fp32 product) against MIOpen's fp32
convolution: error against float64 and time, forward and input gradient, at ResNet-50's stride-1 3x3 shapes (2 x 128 views)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peclr_amd import _capi  # noqa: E402

os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(os.path.dirname(__file__), "..", "..", ".miopen", "db"))
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def timeit(fn, reps=8):
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
    return ts[len(ts) // 2]


SHAPES = [(256, 64, 56), (256, 128, 28), (256, 256, 14), (256, 512, 7)] if len(sys.argv) < 2 else [(8, 128, 9), (5, 256, 6), (6, 64, 10)]
for n, c, hw in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(c + hw)
    x = torch.randn(n, c, hw, hw, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, c, hw, hw, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device="cuda", generator=g) * 0.03).contiguous(memory_format=torch.channels_last)
    w2 = w.permute(0, 2, 3, 1).reshape(c, 9 * c)                 # [Cout][tap][Cin] as it lies in memory
    assert w2.is_contiguous()
    pk = _capi.X6Planes([(w2, False), (w.permute(0, 2, 3, 1).reshape(c * 9, c), 9)]).pack()
    res = {}
    for tr in (128, 256):
        y = _capi.conv3x3_x6p(x, pk.planes[0], c, tile_rows=tr)
        dx = _capi.conv3x3_x6p(gy, pk.planes[1], c, flip=True, tile_rows=tr)
        res[tr] = (y, dx)
    y_mi = F.conv2d(x, w, padding=1)
    dx_mi = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    sub = slice(0, min(n, 6))
    y_ref = F.conv2d(x[sub].double(), w.double(), padding=1)
    dx_ref = torch.ops.aten.convolution_backward(gy[sub].double(), x[sub].double(), w.double(), None, [1, 1], [1, 1], [1, 1], False,
                                                 [0, 0], 1, [True, False, False])[0]
    sy, sd = float(y_ref.abs().max()), float(dx_ref.abs().max())
    print(f"N={n} C={c} {hw}x{hw}: fwd err/scale x6p {float((res[128][0][sub].double() - y_ref).abs().max()) / sy:.2e} "
          f"MIOpen {float((y_mi[sub].double() - y_ref).abs().max()) / sy:.2e} | dgrad x6p {float((res[128][1][sub].double() - dx_ref).abs().max()) / sd:.2e} "
          f"MIOpen {float((dx_mi[sub].double() - dx_ref).abs().max()) / sd:.2e} | 128 == 256 tiles: {torch.equal(res[128][0], res[256][0])} {torch.equal(res[128][1], res[256][1])}")
    fl = 2 * n * hw * hw * 9 * c * c
    t = {tr: (timeit(lambda tr=tr: _capi.conv3x3_x6p(x, pk.planes[0], c, tile_rows=tr)),
              timeit(lambda tr=tr: _capi.conv3x3_x6p(gy, pk.planes[1], c, flip=True, tile_rows=tr))) for tr in (128, 256)}
    tm = (timeit(lambda: F.conv2d(x, w, padding=1)),
          timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])))
    print(f"    fwd: x6p/128 {t[128][0]:7.1f} us ({fl / t[128][0] / 1e6:5.1f} TF)  x6p/256 {t[256][0]:7.1f} us ({fl / t[256][0] / 1e6:5.1f} TF)  MIOpen {tm[0]:7.1f} us ({fl / tm[0] / 1e6:5.1f} TF)"
          f"   dgrad: x6p/128 {t[128][1]:7.1f}  x6p/256 {t[256][1]:7.1f}  MIOpen {tm[1]:7.1f} us", flush=True)
