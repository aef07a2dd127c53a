import sys, torch
sys.path.insert(0, '/root/repo')
from peclr_amd import _capi, bn2d as B, resnet
DEV = 'cuda'
cin, cmid, hw, n = 1024, 256, 14, 256
g = torch.Generator().manual_seed(1)
x0 = (torch.randn(n, cin, hw, hw, generator=g) * 0.7 + 0.3).to(DEV).contiguous(memory_format=torch.channels_last)
res = {}
for fused in (False, True):
    torch.manual_seed(7)
    block = resnet.Bottleneck(cin, cmid, norm_layer=B.FusedBatchNormAct2d).to(DEV).to(memory_format=torch.channels_last).train()
    B.enable_hip_batchnorm(block)
    B._BN_STATS_IN_GEMM = fused
    x = x0.clone().requires_grad_()
    out, identity = B.fork_conv1x1(block.conv1, x, stats_for=block.bn1)
    st = getattr(out, '_peclr_bn_stats', None)
    a1 = resnet._bn(block.bn1, out, relu=True)
    c2 = block.conv2(a1)
    a2 = resnet._bn(block.bn2, c2, relu=True)
    c3 = resnet._conv(block.conv3, a2, block.bn3)
    st3 = getattr(c3, '_peclr_bn_stats', None)
    y = resnet._bn(block.bn3, c3, identity, relu=True)
    res[fused] = dict(out=out.detach(), a1=a1.detach(), c2=c2.detach(), a2=a2.detach(), c3=c3.detach(), y=y.detach(),
                      rm1=block.bn1.running_mean.clone(), rv1=block.bn1.running_var.clone(), rm3=block.bn3.running_mean.clone(), rv3=block.bn3.running_var.clone())
    print(fused, 'stats attached', st is not None, st3 is not None)
for k in res[True]:
    a, b = res[True][k], res[False][k]
    print(k, float((a - b).abs().max()), float(b.abs().max()))
