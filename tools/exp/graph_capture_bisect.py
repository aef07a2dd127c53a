# Experiment: whole-step hipGraph capture of Hybrid2Model (ResNet-18, NHWC), graph vs eager loss curves.
# Result on PyTorch 2.10+rocm7.0 / MI355X: capture + replay is correct with the stock or the fused HIP
# BatchNorm glue and the HIP head/loss kernels as long as gradients are re-allocated inside the capture
# (zero_grad(set_to_none=True)); it goes wrong only when .grad tensors pre-exist as views into the flat
# all-reduce buckets (in-place AccumulateGrad on the leaf stream).  See DESIGN.md section 8.
import sys, copy, warnings
sys.path.insert(0, '/root/repo')
warnings.simplefilter("ignore")
import torch
from peclr_amd import Hybrid2Model, hybrid2_config
from peclr_amd import dist as pdist
from peclr_amd.bn2d import enable_hip_batchnorm
DEV = "cuda:0"; n = 8
torch.manual_seed(11)
cfg = hybrid2_config(resnet_size="18", projection_head_input_dim=512, augmentation=["crop", "rotate"], batch_size=n, num_samples=64, pretrained=False)
g = torch.Generator().manual_seed(12)
batch = {"transformed_image1": torch.randn(n, 3, 64, 64, generator=g), "transformed_image2": torch.randn(n, 3, 64, 64, generator=g),
         "jitter_x_1": torch.randint(-14, 1, (n,), generator=g), "jitter_x_2": torch.randint(-14, 1, (n,), generator=g),
         "jitter_y_1": torch.randint(-14, 1, (n,), generator=g), "jitter_y_2": torch.randint(-14, 1, (n,), generator=g),
         "angle_1": torch.randint(-45, 46, (n,), generator=g).double(), "angle_2": torch.randint(-45, 46, (n,), generator=g).double()}
batch = {k: v.to(DEV) for k, v in batch.items()}
for k in ("transformed_image1", "transformed_image2"): batch[k] = batch[k].contiguous(memory_format=torch.channels_last)
base = Hybrid2Model(cfg).to(DEV).train(); base.encoder = base.encoder.to(memory_format=torch.channels_last)
for variant in ("stock_none", "hipbn_none", "stock_buckets", "stock_none_nostats"):
    def loop(graph):
        m = copy.deepcopy(base)
        if variant.startswith("hipbn"): enable_hip_batchnorm(m.encoder)
        params = [p for nme, p in m.named_parameters() if "final_layer" not in nme]
        opt = torch.optim.SGD(params, lr=0.05)
        red = pdist.GradReducer(m.parameters(), bucket_bytes=1 << 20) if variant.endswith("buckets") else None
        def zero():
            if red: red.zero_grad()
            else: opt.zero_grad(set_to_none=True)
        def fwd():
            if variant.endswith("nostats"):
                z, rs, nn_ = m._project(batch); return m._loss(z, nn_)[0]
            return m.training_step(batch, 0)["loss"]
        def step():
            zero(); loss = fwd(); loss.backward(); opt.step(); return loss
        losses = []
        if not graph: return [float(step()) for _ in range(7)]
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(3): losses.append(float(step()))
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph(); zero()
        with torch.cuda.graph(gr):
            loss = fwd(); loss.backward(); opt.step()
            if red: red.zero_grad()
        for i in range(4): gr.replay(); losses.append(float(loss))
        return losses
    e, gl = loop(False), loop(True)
    print(variant, "eager", [round(v, 4) for v in e]); print(variant, "graph", [round(v, 4) for v in gl])
