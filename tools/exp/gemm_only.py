# Runs only the fused dgrad+add GEMM at the four bottleneck shapes (for PMC passes).
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import _capi
for (r, cmid, cin) in ((802816, 64, 256), (200704, 128, 512), (50176, 256, 1024), (12544, 512, 2048)):
    a, b, d = torch.randn(r, cmid, device="cuda"), torch.randn(cmid, cin, device="cuda"), torch.randn(r, cin, device="cuda")
    for _ in range(3):
        _capi.gemm_add(_capi.GEMM_NN, a, b, d)
torch.cuda.synchronize()
