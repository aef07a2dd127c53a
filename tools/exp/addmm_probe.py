# Experiment: dX = dY W + dRes at the bottleneck-entry shapes: torch.addmm (hipBLASLt/rocBLAS fp32) vs peclr_gemm_add_f32.
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi

def timed(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters

for (r, cmid, cin) in ((802816, 64, 256), (802816, 128, 256), (200704, 128, 512), (200704, 256, 512), (50176, 256, 1024), (50176, 512, 1024), (12544, 512, 2048)):
    a, b, d = torch.randn(r, cmid, device="cuda"), torch.randn(cmid, cin, device="cuda"), torch.randn(r, cin, device="cuda")
    t_lib = timed(lambda: torch.addmm(d, a, b))
    t_inp = timed(lambda: d.addmm_(a, b))
    t_my = timed(lambda: _capi.gemm_add(_capi.GEMM_NN, a, b, d))
    d = torch.randn(r, cin, device="cuda")
    err = float((torch.addmm(d, a, b) - _capi.gemm_add(_capi.GEMM_NN, a, b, d)).abs().max())
    ref = (a[:64].double() @ b.double() + d[:64].double())
    e_lib = float((torch.addmm(d, a, b)[:64].double() - ref).abs().max()); e_my = float((_capi.gemm_add(_capi.GEMM_NN, a, b, d)[:64].double() - ref).abs().max())
    fl = 2 * r * cmid * cin
    print(f"R={r} K={cmid} N={cin}: addmm {t_lib:.0f} us ({fl / t_lib / 1e6:.0f} TF) in-place {t_inp:.0f} us | peclr {t_my:.0f} us ({fl / t_my / 1e6:.0f} TF) | err vs f64: lib {e_lib:.1e} peclr {e_my:.1e}", flush=True)
