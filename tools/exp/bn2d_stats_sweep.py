"""bn2d_stats over ResNet-50's 53 BatchNorm shapes at 2x128 views @224: total time per step for a kernel variant
(PECLR_BN2D_STATS_VARIANT = 10*U + PIPE, set in the environment of THIS process) and n_split policy."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi

shapes = [(256 * 112 * 112, 64)]
for planes, blocks, hw in ((64, 3, 56), (128, 4, 28), (256, 6, 14), (512, 3, 7)):
    for b in range(blocks):
        first_hw = hw * 2 if (b == 0 and planes != 64) else hw
        shapes += [(256 * first_hw * first_hw, planes), (256 * hw * hw, planes), (256 * hw * hw, planes * 4)]
        if b == 0:
            shapes.append((256 * hw * hw, planes * 4))
assert len(shapes) == 53
lib = _capi.lib()
mult = float(os.environ.get("SPLIT_MULT", "1"))
bufs = {}
tot_us, tot_bytes, per = 0.0, 0, []
for r, c in shapes:
    if (r, c) not in bufs:
        bufs[(r, c)] = torch.randn(r, c, device="cuda")
    x = bufs[(r, c)]
    ns = max(1, int(_capi.bn2d_n_split(r, c, 0) * mult))
    part = torch.empty((2 * ns + 1, c), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.peclr_bn2d_stats(x.data_ptr(), 0, r, c, None, part.data_ptr(), ns, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.peclr_bn2d_stats(x.data_ptr(), 0, r, c, None, part.data_ptr(), ns, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    tot_us += us
    tot_bytes += 4 * r * c
    per.append((r, c, ns, round(us, 1), round(4 * r * c / us / 1e6, 2)))
big = [p for p in per if p[0] * p[1] * 4 > 50e6]
print(json.dumps({"variant": os.environ.get("PECLR_BN2D_STATS_VARIANT", "default"), "split_mult": mult,
                  "total_us": round(tot_us, 1), "avg_TBps": round(tot_bytes / tot_us / 1e6, 3),
                  "distinct": sorted(set(per))[:0] or [list(p) for p in sorted(set(per), key=lambda t: -t[0] * t[1])[:12]]}))
