# Fused dgrad+add GEMM at the four bottleneck shapes + two square shapes, timed (PECLR_GEMM_VARIANT in the env picks the kernel).
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi
res = {"variant": os.environ.get("PECLR_GEMM_VARIANT", "0")}
tot = 0.0
for (r, cmid, cin, cnt) in ((802816, 64, 256, 3), (200704, 128, 512, 4), (50176, 256, 1024, 6), (12544, 512, 2048, 3)):
    a, b, d = torch.randn(r, cmid, device="cuda"), torch.randn(cmid, cin, device="cuda"), torch.randn(r, cin, device="cuda")
    for _ in range(3):
        out = _capi.gemm_add(_capi.GEMM_NN, a, b, d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = _capi.gemm_add(_capi.GEMM_NN, a, b, d)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    ref = a[:4096] @ b + d[:4096]
    err = float((out[:4096] - ref).abs().max() / ref.abs().max())
    res[f"{r}x{cmid}x{cin}"] = {"us": round(us, 1), "tf": round(2.0 * r * cmid * cin / us / 1e6, 1), "err": err}
    tot += cnt * us
    del a, b, d, out
res["per_step_us"] = round(tot)
for n in (4096,):
    a, b = torch.randn(n, n, device="cuda"), torch.randn(n, n, device="cuda")
    for layout, name in ((_capi.GEMM_NT, "nt"), (_capi.GEMM_NN, "nn")):
        for _ in range(2):
            _capi.gemm(layout, a, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _capi.gemm(layout, a, b)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 200
        res[f"{name}_{n}"] = {"us": round(us, 1), "tf": round(2.0 * n ** 3 / us / 1e6, 1)}
print(json.dumps(res))
