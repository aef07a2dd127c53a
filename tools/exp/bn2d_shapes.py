# Experiment: the four big bn2d kernels per ResNet-50 activation shape (2x128 views @224, fp32): where is the
# average bandwidth lost?
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi
from peclr_amd.bn2d import FusedBatchNormAct2d

for (hw, c, res) in ((56, 64, False), (56, 256, True), (28, 128, False), (28, 512, True), (14, 256, False), (14, 1024, True), (7, 512, False), (7, 2048, True)):
    bn = FusedBatchNormAct2d(c).cuda().train(); bn.hip = True
    x = torch.randn(256, c, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_()
    r = torch.randn_like(x).requires_grad_() if res else None
    for it in range(8):
        if it == 3: _capi.EVENT_LOG = {}
        y = bn(x, r, True); y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    row = []
    for k, v in _capi.EVENT_LOG.items():
        if "finalize" in k: continue
        us = sum(s.elapsed_time(e) for s, e, *_ in v) / len(v) * 1e3
        row.append(f"{k[5:]} {us:6.1f}us {v[0][2] / us / 1e3:5.0f}GB/s")
    _capi.EVENT_LOG = None
    print(f"{hw:2d}x{hw:<2d} C={c:<4d} res={int(res)} ({256*hw*hw*c*4/1e6:5.0f} MB): " + " | ".join(row), flush=True)
