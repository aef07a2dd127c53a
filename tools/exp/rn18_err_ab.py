"""ResNet-18 fp32 in-tree vs float64 stock: which routing switch carries the input-gradient error?"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import bn2d as B
from peclr_amd.config import Config
from peclr_amd.encoder import get_wrapper_model
DEV = "cuda:0"
torch.manual_seed(5)
net0 = get_wrapper_model(Config({"resnet_size": "18"}), False).to(DEV).to(memory_format=torch.channels_last).train()
for p in net0.final_layer.parameters():
    p.requires_grad_(False)
ref = copy.deepcopy(net0); B.enable_hip_batchnorm(ref, False); ref = ref.double()
g = torch.Generator().manual_seed(128)
x = torch.randn(16, 3, 128, 128, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
gy = (torch.randn(16, 512, generator=g) / 512).to(DEV)
def run(net, x, gy):
    x = x.clone().requires_grad_(); y = net(x); y.backward(gy.to(y.dtype)); torch.cuda.synchronize()
    return y.detach().double(), x.grad.double()
want = run(ref, x.double(), gy.double())
rel = lambda a, b: float((a - b).norm() / b.norm())
stock = copy.deepcopy(net0); B.enable_hip_batchnorm(stock, False)
r = run(stock, x, gy); print("stock fp32      ", rel(r[0], want[0]), rel(r[1], want[1]), flush=True)
for name, kw in (("force all", dict(force=True)), ("no force", {}), ("force, no stats fuse", dict(force=True, bn_stats_in_gemm=False)),
                 ("force, no bwd fuse", dict(force=True, bn_bwd_in_gemm=False)), ("force, no 3x3", dict(force=True, conv3x3_x6=False)),
                 ("force, no s2", dict(force=True, conv_s2_x6=False, conv_s2_dgrad_x6=False)), ("force, no x6", dict(force=True, gemm_x6=False)),
                 ("force, no 3x3 wgrad", dict(force=True, conv3x3_wgrad_x6=False))):
    net = copy.deepcopy(net0); B.enable_hip_batchnorm(net)
    with B.routing(**kw):
        r = run(net, x, gy); B.end_backward()
    print(f"{name:22s}", rel(r[0], want[0]), rel(r[1], want[1]), flush=True)
