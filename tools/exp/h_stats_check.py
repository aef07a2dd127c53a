import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi as capi
DEV = "cuda:0"; dt = torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(0)
n, cin, cout, hw = 16, 256, 64, 56
x = (torch.randn(n, cin, hw, hw, device=DEV, generator=g) * 0.7 + 0.3).to(dt).contiguous(memory_format=torch.channels_last)
w = torch.randn(cout, cin, device=DEV, generator=g) * 0.05
pk = capi.HPlanes([(w, False)], dt).pack()
x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)
gamma, beta = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
for shift in (torch.zeros(cout, device=DEV), torch.randn(cout, device=DEV, generator=g) * 0.3):
    y, partial, ns = capi.gemm_h(x2, pk.planes[0], cout, stat_shift=shift)
    y4 = y.view(n, hw, hw, cout).permute(0, 3, 1, 2)
    rm, rv, nbt = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV), torch.zeros((), device=DEV, dtype=torch.int64)
    save_a, ss_a = capi._bn2d_scale_shift(y4, gamma, beta, rm.clone(), rv.clone(), nbt.clone(), True, 1e-5, 0.1, pre=(partial, ns, shift))
    save_b, ss_b = capi._bn2d_scale_shift(y4, gamma, beta, rm.clone(), rv.clone(), nbt.clone(), True, 1e-5, 0.1)
    yd = y.double()
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    inv = 1 / (var + 1e-5).sqrt()
    for nm, s in (("fused", save_a), ("separate", save_b)):
        print(nm, "mean err", float((s[0].double() - mean).abs().max()), "invstd rel err", float(((s[1].double() - inv) / inv).abs().max()))
