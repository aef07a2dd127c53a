"""Per-shape time of the fp32 1x1 forward products of ResNet-50 at C2 (2 x 128 views @224), statistics epilogue on: algorithmic
bytes / time next to the six-product MFMA time at 2.5 PFLOP/s.  python tools/exp/x6p_shapes_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
from peclr_amd import _capi as capi

DEV = "cuda:0"
junk = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)


def timed(fn, reps=6):
    ts = []
    for r in range(reps):
        junk.fill_(r)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        if r >= 2:
            ts.append(s.elapsed_time(e) * 1e3)
    return sum(ts) / len(ts)


B = 256
shapes = []
for name, hw, cin, mid in (("layer1", 56, 256, 64), ("layer2", 28, 512, 128), ("layer3", 14, 1024, 256), ("layer4", 7, 2048, 512)):
    m = B * hw * hw
    shapes.append((f"{name} conv1 {cin}->{mid}", m, mid, cin))
    shapes.append((f"{name} conv3 {mid}->{4 * mid}", m, 4 * mid, mid))
print(f"{'shape':28s} {'M':>8s} {'N':>5s} {'K':>5s} {'us':>8s} {'TB/s':>6s} {'mfma us':>8s} {'hbm us @5.8':>11s}")
for name, m, n, k in shapes:
    a = torch.randn(m, k, device=DEV)
    bt = torch.randn(n, k, device=DEV) * 0.05
    planes = capi.X6Planes([(bt, False)]).pack().planes[0]
    shift = torch.zeros(n, device=DEV)
    for tr in (128, 256):
        us = timed(lambda: capi.gemm_x6p(a, planes, n, tile_rows=tr, stat_shift=shift))
        nbytes = 4.0 * m * (k + n)
        print(f"{name:28s} {m:8d} {n:5d} {k:5d} {us:8.1f} {nbytes / us / 1e6:6.2f} {12.0 * m * n * k / 2.5e15 * 1e6:8.1f} {nbytes / 5.8e12 * 1e6:11.1f}   tile_rows {tr}")
