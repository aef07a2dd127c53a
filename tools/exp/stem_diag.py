# Diagnostic: how far apart are conv1 weight gradients between (a) two identical stock runs, (b) stock vs fused
# glue without the pooled stem, (c) stock vs fused with the pooled stem?
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peclr_amd import hybrid2_config
from peclr_amd.bn2d import enable_hip_batchnorm
from peclr_amd.encoder import get_wrapper_model
torch.manual_seed(1)
cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, pretrained=False)
base = get_wrapper_model(cfg, False).cuda().to(memory_format=torch.channels_last).train()
x = torch.randn(8, 3, 96, 96, device="cuda").contiguous(memory_format=torch.channels_last)
def run(kind):
    m = copy.deepcopy(base)
    if kind != "stock":
        enable_hip_batchnorm(m)
    if kind == "fused_nopool":
        m.features[1].default_pool = False
        m.features[3] = torch.nn.MaxPool2d(3, 2, 1)
    xx = x.clone().requires_grad_()
    y = m(xx); y.square().mean().backward()
    return y.detach(), m.features[0].weight.grad.clone(), xx.grad.clone()
ref = run("stock")
for kind in ("stock", "fused_nopool", "fused"):
    y, g, dx = run(kind)
    print(kind, "y", float((y - ref[0]).abs().max()), "conv1.wgrad", float((g - ref[1]).abs().max()), "/", float(ref[1].abs().max()),
          "dx_in", float((dx - ref[2]).abs().max()), "/", float(ref[2].abs().max()))
