import sys, torch
sys.path.insert(0, '/root/repo')
from peclr_amd import _capi, bn2d as B, resnet
DEV = 'cuda'
cin, cmid, hw, n = 1024, 256, 14, 256
g = torch.Generator().manual_seed(1)
x0 = (torch.randn(n, cin, hw, hw, generator=g) * 0.7 + 0.3).to(DEV).contiguous(memory_format=torch.channels_last)
gy = torch.randn(n, cin, hw, hw, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
res = {}
for key, fused in (("u0", False), ("f0", True), ("u1", False), ("f1", True)):
    torch.manual_seed(7)
    block = resnet.Bottleneck(cin, cmid, norm_layer=B.FusedBatchNormAct2d).to(DEV).to(memory_format=torch.channels_last).train()
    B.enable_hip_batchnorm(block)
    B._BN_STATS_IN_GEMM = fused
    x = x0.clone().requires_grad_()
    y = block(x)
    y.backward(gy)
    res[key] = dict(y=y.detach().clone(), dx=x.grad.clone(), **{n_: p.grad.clone() for n_, p in block.named_parameters()})
for a, b in (("u0", "u1"), ("f0", "f1"), ("f0", "u0")):
    print(a, b, {k: (float((res[a][k] - res[b][k]).abs().max()), float(res[b][k].abs().max())) for k in res[a]})
