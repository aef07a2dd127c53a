# Experiment: the head (K1..K7) and NT-Xent operators replay correctly from a captured hipGraph
# (graph vs eager loss curves are identical, torch SGD).  Whole-step capture with the ResNet backbone
# + flat gradient buckets still produced wrong gradients in the last residual block and is NOT shipped
# (DESIGN.md section 8).
import sys, copy, warnings
sys.path.insert(0, '/root/repo')
warnings.simplefilter("ignore")
import torch
from peclr_amd import ops
DEV = "cuda:0"
n = 16
torch.manual_seed(0)
class Net(torch.nn.Module):
    def __init__(self, mode):
        super().__init__()
        self.mode = mode
        self.enc = torch.nn.Linear(96, 64)
        self.l1 = torch.nn.Linear(64, 128); self.bn = torch.nn.BatchNorm1d(128); self.l2 = torch.nn.Linear(128, 128, bias=False)
    def forward(self, x, spec):
        h = torch.tanh(self.enc(x))
        if self.mode in ("head", "both"):
            st = ops.BNState(True, self.bn.eps, 0.1, self.bn.running_mean, self.bn.running_var, self.bn.num_batches_tracked)
            z, rs = ops.head_align(h, self.l1.weight, self.l1.bias, self.bn.weight, self.bn.bias, self.l2.weight, st, spec)
        else:
            z = torch.nn.functional.normalize(self.l2(torch.relu(self.bn(self.l1(h)))))
        if self.mode in ("loss", "both"):
            return ops.ntxent(z, n, 0.5)[0]
        s = torch.exp(z @ z.t() / 0.5); s = s - torch.diag(torch.diag(s))
        pos = torch.exp((z[:n] * z[n:]).sum(-1) / 0.5); pos = torch.cat([pos, pos])
        return -torch.log(pos / s.sum(-1)).mean()
x = torch.randn(2 * n, 96, device=DEV)
g = torch.Generator().manual_seed(1)
spec = ops.AlignSpec(n_pairs=n, crop=True, rotate=True,
                     jitter=tuple(torch.randint(-14, 1, (n,), generator=g).to(DEV) for _ in range(4)), extents=(224.0, 224.0),
                     angles=tuple(torch.randint(-45, 46, (n,), generator=g).double().to(DEV) for _ in range(2)))
for mode in ("none", "head", "loss", "both"):
    base = Net(mode).to(DEV).train()
    def loop(graph):
        m = copy.deepcopy(base); opt = torch.optim.SGD(m.parameters(), lr=0.5); losses = []
        def step():
            opt.zero_grad(set_to_none=True); loss = m(x, spec); loss.backward(); opt.step(); return loss
        if not graph:
            return [float(step()) for _ in range(8)]
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(3): losses.append(float(step()))
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph(); opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(gr):
            loss = m(x, spec); loss.backward(); opt.step()
        for i in range(5):
            gr.replay(); losses.append(float(loss))
        return losses
    e, gl = loop(False), loop(True)
    print(mode, "eager", [round(v, 4) for v in e]); print(mode, "graph", [round(v, 4) for v in gl])
