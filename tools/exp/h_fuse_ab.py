"""16-bit: BatchNorm backward reduction fused into the dgrad epilogues vs the separate pass, arm against arm."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import bn2d as B, resnet
DEV = "cuda:0"
dtype = torch.bfloat16
for cin, cmid, hw, n in ((256, 64, 56, 16), (1024, 256, 14, 64)):
    g = torch.Generator().manual_seed(1)
    x0 = (torch.randn(n, cin, hw, hw, generator=g) * 0.7 + 0.3).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, cin, hw, hw, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    res = {}
    for name, kw in (("all", {}), ("no_bwd_fuse", dict(bn_bwd_in_gemm=False)), ("no_stats_fuse", dict(bn_stats_in_gemm=False)),
                     ("no_lazy", dict(lazy_residual_grad=False)), ("no_wgrad16", dict(wgrad16=False)), ("no_conv16", dict(conv16=False))):
        torch.manual_seed(7)
        net = torch.nn.Sequential(resnet.Bottleneck(cin, cmid, norm_layer=B.FusedBatchNormAct2d),
                                  resnet.Bottleneck(cin, cmid, norm_layer=B.FusedBatchNormAct2d)).to(DEV).to(memory_format=torch.channels_last).train()
        B.enable_hip_batchnorm(net)
        with B.routing(force=True, **kw):
            x = x0.clone().requires_grad_()
            with torch.autocast("cuda", dtype=dtype):
                y = net(x)
            y.backward(gy)
            torch.cuda.synchronize()
            B.end_backward()
        res[name] = (y.detach().float(), x.grad.float(), {k: p.grad.float() for k, p in net.named_parameters()})
    ref = res["all"]
    for name, r in res.items():
        rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
        worst = max(((rel(r[2][k], ref[2][k]), k) for k in ref[2]))
        print(f"[{cin} {hw}] {name:14s} y {rel(r[0], ref[0]):.2e} dx {rel(r[1], ref[1]):.2e} worst grad {worst[0]:.2e} {worst[1]}", flush=True)
