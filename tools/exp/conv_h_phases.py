"""Shader clocks per phase of conv_h_kernel (workgroup 8, thread 0) at ResNet-50's 1x1 shapes: needs the timing build,
PECLR_HIP_LIB=tools/exp/ab/libtiming.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi as capi
DEV = "cuda:0"; dt = torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(0)
L = capi.lib()
buf = (ctypes.c_ulonglong * 8)()
def phases(what, fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); L.peclr_debug_conv_h_timing(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); L.peclr_debug_conv_h_timing(buf, 1)
    us = a.elapsed_time(b) / reps * 1e3
    p = [buf[k] / reps for k in range(4)]
    print(f"{what:44s} {us:7.1f} us | workgroup 8: set-up {p[0]:6.0f}  main loop {p[1]:6.0f}  statistics {p[2]:6.0f}  stores {p[3]:6.0f}  total {sum(p):6.0f} clocks")
for r, cmid in ((256 * 56 * 56, 64), (256 * 28 * 28, 128), (256 * 14 * 14, 256), (256 * 7 * 7, 512)):
    cin = 4 * cmid
    x = torch.randn(r, cin, device=DEV, generator=g).to(dt)
    y1 = torch.randn(r, cmid, device=DEV, generator=g).to(dt)
    w1 = torch.randn(cmid, cin, device=DEV, generator=g) * 0.05
    pk = capi.HPlanes([(w1, False), (w1, True), (w1.t().contiguous(), False)], dt).pack()
    shift, shift4 = torch.zeros(cmid, device=DEV), torch.zeros(cin, device=DEV)
    mean, invstd = torch.zeros(cin, device=DEV), torch.ones(cin, device=DEV)
    save, ss = torch.stack([mean, invstd]).contiguous(), torch.stack([invstd, mean]).contiguous()
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (r, cin // 32), device=DEV, dtype=torch.int32)
    phases(f"conv1 fwd [{r}, {cin}] -> {cmid} + stats", lambda: capi.gemm_h(x, pk.planes[0], cmid, stat_shift=shift))
    phases(f"conv3 fwd [{r}, {cmid}] -> {cin} + stats", lambda: capi.gemm_h(y1, pk.planes[2], cin, stat_shift=shift4))
    phases(f"dgrad [{r}, {cmid}] -> {cin} plain", lambda: capi.gemm_h(y1, pk.planes[1], cin))
    phases(f"fork dgrad [{r}, {cmid}] -> {cin}", lambda: capi.gemm_h(y1, pk.planes[1], cin, x, addend_mask=mask, bn_bwd=(x, save, ss, mask, True)))
