import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi as capi
DEV = "cuda:0"; dt = torch.bfloat16
def nhwc(t): return t.contiguous(memory_format=torch.channels_last)
def where(bad, name):
    idx = bad.nonzero()
    print(f"  {name}: {idx.shape[0]} bad")
    if idx.shape[0]:
        for d, nm in enumerate("nchw"):
            u, c = idx[:, d].unique(return_counts=True)
            print(f"    {nm}: {u.tolist()[:20]} counts {c.tolist()[:20]}")
for nb, cin, cout, h, w in ((3, 128, 128, 14, 14), (1, 64, 128, 5, 62), (2, 128, 64, 1, 1)):
    g = torch.Generator(device=DEV).manual_seed(1)
    x = nhwc(torch.randn(nb, cin, h, w, device=DEV, generator=g).to(dt))
    wt = nhwc(torch.randn(cout, cin, 3, 3, device=DEV, generator=g) * 0.05)
    w4 = wt.permute(0, 2, 3, 1)
    pk = capi.HPlanes([(w4.reshape(cout, 9 * cin), False), (w4.reshape(cout * 9, cin), 9)], dt).pack()
    print(nb, cin, cout, h, w)
    y_ref = capi.conv_h(x, pk.planes[0], cout, tile_rows=128)
    y = capi.conv_h(x, pk.planes[0], cout, tile_rows=1)
    where((y.float() - y_ref.float()).abs() > 0.02 * y_ref.float().abs().max(), "forward ring vs per-tap")
    shift = torch.zeros(cout, device=DEV)
    y2, part, ns = capi.conv_h(x, pk.planes[0], cout, tile_rows=1, stat_shift=shift)
    where((y2.float() - y_ref.float()).abs() > 0.02 * y_ref.float().abs().max(), "forward ring + stats vs per-tap")
    where(~torch.isfinite(y2.float()), "non-finite")
    gy = nhwc(torch.randn(nb, cout, h, w, device=DEV, generator=g).to(dt))
    d_ref = capi.conv_h(gy, pk.planes[1], cin, flip=True, tile_rows=128)
    d = capi.conv_h(gy, pk.planes[1], cin, flip=True, tile_rows=1)
    where((d.float() - d_ref.float()).abs() > 0.02 * d_ref.float().abs().max(), "input gradient ring vs per-tap")
