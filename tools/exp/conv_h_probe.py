"""Per-launch time of the 16-bit convolution kernels at ResNet-50's shapes (2 x 128 views @224): us and algorithmic TB/s.
   python tools/exp/conv_h_probe.py [tile_rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from peclr_amd import _capi as capi

DEV = "cuda:0"
dt = torch.bfloat16
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


g = torch.Generator(device=DEV).manual_seed(0)
print(f"tile_rows {tile}, narrow {os.environ.get('PECLR_CONV_H_NARROW', '0')}")
for r, cmid in ((256 * 56 * 56, 64), (256 * 28 * 28, 128), (256 * 14 * 14, 256), (256 * 7 * 7, 512)):
    cin = 4 * cmid
    x = torch.randn(r, cin, device=DEV, generator=g).to(dt)
    y1 = torch.randn(r, cmid, device=DEV, generator=g).to(dt)
    w1 = torch.randn(cmid, cin, device=DEV, generator=g) * 0.05
    pk = capi.HPlanes([(w1, False), (w1, True), (w1.t().contiguous(), False)], dt).pack()
    shift = torch.zeros(cmid, device=DEV)
    shift4 = torch.zeros(cin, device=DEV)
    mean, invstd = torch.zeros(cin, device=DEV), torch.ones(cin, device=DEV)
    save, ss = torch.stack([mean, invstd]).contiguous(), torch.stack([invstd, mean]).contiguous()
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (r, cin // 32), device=DEV, dtype=torch.int32)
    t = timeit(lambda: capi.gemm_h(x, pk.planes[0], cmid, stat_shift=shift, tile_rows=tile))
    by = 2 * r * (cin + cmid)
    print(f"conv1 fwd  [{r}, {cin}] -> {cmid} + stats      : {t:7.1f} us  {by / t / 1e6:5.2f} TB/s")
    t = timeit(lambda: capi.gemm_h(y1, pk.planes[2], cin, stat_shift=shift4, tile_rows=tile))
    print(f"conv3 fwd  [{r}, {cmid}] -> {cin} + stats      : {t:7.1f} us  {by / t / 1e6:5.2f} TB/s")
    t = timeit(lambda: capi.gemm_h(y1, pk.planes[1], cin, tile_rows=tile))
    print(f"dgrad      [{r}, {cmid}] -> {cin} plain        : {t:7.1f} us  {by / t / 1e6:5.2f} TB/s")
    t = timeit(lambda: capi.gemm_h(y1, pk.planes[1], cin, x, tile_rows=tile))
    by2 = 2 * r * (2 * cin + cmid)
    print(f"dgrad      [{r}, {cmid}] -> {cin} + addend     : {t:7.1f} us  {by2 / t / 1e6:5.2f} TB/s")
    t = timeit(lambda: capi.gemm_h(y1, pk.planes[1], cin, x, addend_mask=mask, bn_bwd=(x, save, ss, mask, True), tile_rows=tile))
    by3 = 2 * r * (3 * cin + cmid) + r * cin // 8
    print(f"fork dgrad [{r}, {cmid}] -> {cin} + mask add + bn : {t:7.1f} us  {by3 / t / 1e6:5.2f} TB/s")
    t = timeit(lambda: x.clone())
    print(f"           clone of [{r}, {cin}]                  : {t:7.1f} us  {4 * r * cin / t / 1e6:5.2f} TB/s")
