"""Time peclr_gemm_add_bf16 / _f16 at ResNet-50's four bottleneck-entry shapes (2 x 128 views @224).
Run twice: as is (128 x 128-tile kernel) and with PECLR_GEMM_TILE=64 (round-1 kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peclr_amd import _capi  # noqa: E402

SHAPES = [(256 * 56 * 56, 256, 64), (256 * 28 * 28, 512, 128), (256 * 14 * 14, 1024, 256), (256 * 7 * 7, 2048, 512)]
for half in (torch.bfloat16, torch.float16):
    for m, n, k in SHAPES:
        a = torch.randn(m, k, device="cuda").to(half)
        bt = torch.randn(n, k, device="cuda").to(half)
        d = torch.randn(m, n, device="cuda").to(half)
        junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
        ts = []
        for it in range(12):
            junk.zero_()                       # evict the operands from the Infinity Cache, as in the step
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            _capi.gemm_add_half(a, bt, d)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        ts = sorted(ts[2:])
        nbytes = 2 * (m * k + n * k + 2 * m * n)
        us = ts[len(ts) // 2]
        print(f"{str(half):16s} M={m:7d} N={n:5d} K={k:4d}  {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s  {2 * m * n * k / us / 1e6:7.1f} TFLOP/s"
              f"  tile={os.environ.get('PECLR_GEMM_TILE', '128')}")
