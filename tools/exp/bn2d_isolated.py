"""The four streaming BatchNorm kernels in ISOLATION (no convolution before them, so no dirty lines of a producer in the
Infinity Cache) at ResNet-50's large shapes, 2x128 views: achieved TB/s from HIP events around each launch."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi

res = {}
for (n, c, hw) in ((256, 64, 112), (256, 256, 56), (256, 64, 56), (256, 512, 28), (256, 1024, 14)):
    x = torch.randn(n, c, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    r = torch.randn(n, c, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, c, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    rm, rv, nbt = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"), torch.zeros((), device="cuda", dtype=torch.int64)
    for _ in range(2):
        y, save, ss, mask = _capi.bn2d_fwd(x, r, g, b, rm, rv, nbt, True, 1e-5, 0.1, relu=True, want_mask=True)
        _capi.bn2d_bwd(dy, x, None, mask, save, ss, True, True, True)
    torch.cuda.synchronize()
    _capi.EVENT_LOG = {}
    for _ in range(5):
        y, save, ss, mask = _capi.bn2d_fwd(x, r, g, b, rm, rv, nbt, True, 1e-5, 0.1, relu=True, want_mask=True)
        _capi.bn2d_bwd(dy, x, None, mask, save, ss, True, True, True)
    torch.cuda.synchronize()
    log, _capi.EVENT_LOG = _capi.EVENT_LOG, None
    row = {}
    for name, recs in log.items():
        if "finalize" in name:
            continue
        us = 1e3 * sum(e[0].elapsed_time(e[1]) for e in recs) / len(recs)
        row[name] = {"us": round(us, 1), "TBps": round(recs[0][2] / us / 1e6, 2)}
    res[f"{n}x{c}x{hw}x{hw} ({4 * n * c * hw * hw / 1e6:.0f} MB)"] = row
    del x, r, dy, y
print(json.dumps(res, indent=1))
