"""How does the pair split round?  hi must be fp16(x s) to NEAREST (v_cvt_pk_f16_f32 under the default rounding mode); a truncating
conversion would give every lo the sign of x and the dropped lo.lo products a systematic sign."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from peclr_amd import _capi as capi  # noqa: E402

DEV = "cuda:0"
# one matrix [128, 16]: column values chosen around fp16 rounding boundaries; the maximum 2^14 makes the scale exactly 1
vals = np.array([16384.0, 1.0 + 0.75 * 2 ** -10, 1.0 + 0.25 * 2 ** -10, -(1.0 + 0.75 * 2 ** -10), 1.0 + 0.5 * 2 ** -10, 1.0 + 1.5 * 2 ** -10,
                 3.0 + 2 ** -9 * 0.9, 100.0 + 0.03, 0.1, -0.3, 1e-3, 5e-5, 2.0 ** -14 * 0.75, 2.0 ** -16, 0.0, 7.0], dtype=np.float32)
w = torch.tensor(np.tile(vals, (128, 1)), device=DEV)
pp = capi.X6Planes([(w, False)], pair=True).pack()
torch.cuda.synchronize()
s = float(pp.scales[0])
raw = pp.planes[0].cpu().numpy().view(np.float16)        # chunk: [32-column block][plane] pieces of 512 halves: lane l = column l & 31, k-half l >> 5
hi = np.concatenate([raw[0:8], raw[256:264]])            # column 0, k 0..7 and 8..15 of plane 0
lo = np.concatenate([raw[512:520], raw[512 + 256:512 + 264]])
print("scale", s)
for v, h, l in zip(vals, hi, lo):
    want = np.float16(v * s)
    print(f"x = {v!r:>22}: hi {float(h)!r:>12} (nearest: {float(want)!r:>12}) lo {float(l)!r:>14}  residual after hi + lo: {float(np.float64(v) * s - float(h) - float(l)):.3e}")
assert all(np.float16(v * s) == h for v, h in zip(vals, hi)), "hi is not round-to-nearest"
print("hi = round-to-nearest(x s) for every sample")
