import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi as capi
DEV = "cuda:0"
dtype = torch.bfloat16
for (m, n, k) in [(256, 128, 64), (2352, 128, 512)]:
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn(m, k, device=DEV, generator=g).to(dtype)
    w = torch.randn(n, k, device=DEV, generator=g) * 0.05
    pk = capi.HPlanes([(w, False)], dtype).pack()
    xb = torch.randn(m, n, device=DEV, generator=g).to(dtype)
    mean, invstd = xb.float().mean(0), 1.0 / (xb.float().var(0, unbiased=False) + 1e-5).sqrt()
    ss = torch.stack([invstd, -mean * invstd]).contiguous()
    save = torch.stack([mean, invstd]).contiguous()
    torch.cuda.synchronize()
    for nm, t in (("a", a), ("planes", pk.planes[0]), ("xb", xb), ("save", save), ("ss", ss)):
        print(nm, hex(t.data_ptr()), t.numel() * t.element_size(), flush=True)
    for tr in (256, 128):
        for relu in (False, True):
            print("launch", m, n, k, tr, relu, flush=True)
            dy, partial, ns = capi.gemm_h(a, pk.planes[0], n, bn_bwd=(xb, save, ss, None, relu), tile_rows=tr)
            torch.cuda.synchronize()
            print("ok", float(partial.sum()), flush=True)
