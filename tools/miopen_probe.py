"""Diagnostic: how long does the PyTorch-ROCm/MIOpen backbone take to get going on a fresh box
(kernel JIT / find), and what is the steady-state step?  Usage: python tools/miopen_probe.py [benchmark 0/1] [dtype] [cl 0/1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peclr_amd import resnet

bench = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
cl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bs = int(sys.argv[4]) if len(sys.argv) > 4 else 256
torch.backends.cudnn.benchmark = bool(bench)
dev = torch.device("cuda:0")
m = resnet.resnet50().to(dev).train()
x = torch.randn(bs, 3, 224, 224, device=dev)
if cl:
    m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
opt = torch.optim.SGD(m.parameters(), lr=0.0)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
        y = m(x)
    y.float().sum().backward()
    opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
for i in range(3):
    t = time.time(); step(); print(f"[bench={bench} {dtype} cl={cl} bs={bs}] call {i}: {time.time()-t:.2f} s", flush=True)
t = time.time()
for i in range(5): step()
print(f"[bench={bench} {dtype} cl={cl} bs={bs}] steady: {(time.time()-t)/5*1e3:.1f} ms/step -> {bs*5/(time.time()-t):.0f} img/s", flush=True)
