# Experiment: the fused stem (BN + ReLU + max-pool) kernels at the benchmark shape, per-kernel HIP-event times.
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi
from peclr_amd.bn2d import FusedBatchNormAct2d

for dtype in (torch.float32, torch.bfloat16):
    bn = FusedBatchNormAct2d(64).cuda().train()
    bn.hip = bn.default_relu = bn.default_pool = True
    x = torch.randn(256, 64, 112, 112, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    for it in range(6):
        if it == 3:
            _capi.EVENT_LOG = {}
        y = bn(x)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    for k, v in _capi.EVENT_LOG.items():
        us = sum(s.elapsed_time(e) for s, e, *_ in v) / len(v) * 1e3
        print(dtype, k, round(us, 1), "us", round(v[0][2] / us / 1e3, 1), "GB/s")
    _capi.EVENT_LOG = None
