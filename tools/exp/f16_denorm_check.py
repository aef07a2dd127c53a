"""Does the fp16 matrix-core path keep subnormal inputs?  (MI200's MFMA flushed them; the pair arithmetic relies on `lo` parts below
2^-14 surviving.)  1) rocBLAS / hipBLASLt half GEMM on subnormal inputs; 2) the pair kernel itself: a tensor whose `lo` parts are
all subnormal, against float64."""
import sys

import torch

sys.path.insert(0, ".")
from peclr_amd import _capi as capi  # noqa: E402

DEV = "cuda:0"
a = torch.full((256, 256), 2.0 ** -20, dtype=torch.half, device=DEV)       # subnormal in fp16 (min normal 2^-14)
b = torch.full((256, 256), 1024.0, dtype=torch.half, device=DEV)
print("half GEMM of subnormal inputs: got", float((a @ b)[0, 0]), "expected", 256 * 2.0 ** -10)

# pair kernel: rows of magnitude 2^-20 under a tensor maximum of 1: x s ~ 2^-6, hi ulp 2^-16, lo < 2^-17: subnormal
g = torch.Generator().manual_seed(0)
m, n, k = 4096, 256, 512
x = torch.randn(m, k, generator=g) * (2.0 ** -20)
x[0, 0] = 1.0                                                              # the tensor's maximum
x = x.to(DEV)
bt = (torch.randn(n, k, generator=g) * 0.05).to(DEV)
pp = capi.X6Planes([(bt, False)], pair=True).pack()
out = capi.gemm_x6p(x, pp.planes[0], n, pair=(x.abs().max().reshape(1), pp.scale(0)))
ref = x.double() @ bt.double().t()
rows = slice(1, None)                                                      # the small rows
rel = float(((out.double() - ref)[rows].abs().max()) / ref[rows].abs().max())
print(f"pair GEMM, rows 2^-20 below the tensor maximum: max err / max |ref| over those rows = {rel:.2e}  (2^-11 = {2.0 ** -11:.1e} would mean "
      f"the subnormal lo parts are lost; ~1e-5 = they count)")
