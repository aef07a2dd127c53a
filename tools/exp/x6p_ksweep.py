"""Time of peclr_gemm_x6p_f32 against K at fixed M, N: the slope is the cost of a k-step, the intercept what a tile pays around its loop.
python tools/exp/x6p_ksweep.py"""
import sys
import torch
sys.path.insert(0, ".")
from peclr_amd import _capi as capi

DEV = "cuda:0"
junk = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)


def timed(fn, reps=7):
    ts = []
    for r in range(reps):
        junk.fill_(r)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        if r >= 2:
            ts.append(s.elapsed_time(e) * 1e3)
    return sum(ts) / len(ts)


for m, n in ((50176, 1024), (12544, 2048), (50176, 256), (200704, 512)):
    print(f"M = {m}, N = {n}: tiles of 128 rows: {(m + 127) // 128 * (n // 128)} workgroups")
    prev = None
    for k in (64, 128, 256, 512, 1024, 2048):
        a = torch.randn(m, k, device=DEV)
        bt = torch.randn(n, k, device=DEV) * 0.05
        planes = capi.X6Planes([(bt, False)]).pack().planes[0]
        row = []
        for tr in (128, 256):
            us = timed(lambda: capi.gemm_x6p(a, planes, n, tile_rows=tr))
            row.append(us)
        mf = 12.0 * m * n * k / 2.5e15 * 1e6
        hb = 4.0 * m * (k + n) / 5.8e12 * 1e6
        slope = "" if prev is None else f"   d/dK: {(row[0] - prev[0]) / (k - prev[1]) * 16 * 1e3:7.1f} ns per k-step (128)   mfma share of the slope {(mf - prev[2]) / (row[0] - prev[0]):.2f}"
        print(f"  K = {k:5d}: {row[0]:8.1f} us (128-row tiles) {row[1]:8.1f} us (256)   products {mf:7.1f} us, traffic {hb:6.1f} us{slope}")
        prev = (row[0], k, mf)
