"""MIOpen's fp32 NHWC 1x1 convolutions of ResNet-50 (2 x 128 views @224): forward and input-gradient time per shape,
next to peclr_gemm_x6_f32 on the same GEMM."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: F401,E402  (sets MIOPEN_USER_DB_PATH to the in-tree databases)
from peclr_amd import _capi  # noqa: E402

N = 256
SHAPES = [(256, 64, 56), (64, 256, 56), (512, 128, 28), (128, 512, 28), (1024, 256, 14), (256, 1024, 14), (2048, 512, 7), (512, 2048, 7)]


def timeit(fn, reps=10):
    junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
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


for cin, cout, hw in SHAPES:
    x = torch.randn(N, cin, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, cout, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    r = N * hw * hw
    x2, gy2, w2 = x.permute(0, 2, 3, 1).reshape(r, cin), gy.permute(0, 2, 3, 1).reshape(r, cout), w.reshape(cout, cin)
    wt = w2.t().contiguous()
    t_fwd = timeit(lambda: F.conv2d(x, w))
    t_bwd = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (True, False, False)))
    t_wrw = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False)))
    t6_fwd = timeit(lambda: _capi.gemm_x6(x2, w2))            # y[R,Cout] = x[R,Cin] . W[Cout,Cin]^T
    t6_bwd = timeit(lambda: _capi.gemm_x6(gy2, wt))           # dx[R,Cin] = gy[R,Cout] . Wt[Cin,Cout]^T
    t6_wrw = timeit(lambda: _capi.gemm_x6_tn(gy2, x2))        # dW[Cout,Cin] = gy[R,Cout]^T . x[R,Cin]
    dw_ref = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False))[1]
    dw_err = float((_capi.gemm_x6_tn(gy2, x2) - dw_ref.reshape(cout, cin)).abs().max()) / float(dw_ref.abs().max())
    fl = 2 * r * cin * cout
    print(f"{cin:5d}->{cout:5d} @{hw:2d}  MIOpen fwd {t_fwd:6.1f} ({fl / t_fwd / 1e6:5.0f} TF)  dgrad {t_bwd:6.1f}  wgrad {t_wrw:6.1f} | x6 fwd {t6_fwd:6.1f}  dgrad {t6_bwd:6.1f}  wgrad {t6_wrw:6.1f} (vs MIOpen {dw_err:.1e})")
