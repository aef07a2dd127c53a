import sys, time, argparse, torch
sys.path.insert(0, "/root/repo"); sys.argv=["bench.py"]
import bench
import numpy as np
a = argparse.Namespace(resnet="18", accum=1, size=224, cpu_pairs=8, pairs=32, dtype="fp32")
for thr in (8, 16, 32, 64, 128):
    torch.set_num_threads(thr)
    model = bench.build_model(a, torch.device("cpu"), 32)
    x = torch.randn(64, 3, 224, 224)
    def step():
        model.zero_grad(set_to_none=True)
        h = model.encoder(x); h.square().mean().backward()
    step()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    a50 = argparse.Namespace(resnet="50", accum=1, size=224, cpu_pairs=8, pairs=8, dtype="fp32")
    m50 = bench.build_model(a50, torch.device("cpu"), 8)
    x50 = torch.randn(16, 3, 224, 224)
    def step50():
        m50.zero_grad(set_to_none=True)
        h = m50.encoder(x50); h.square().mean().backward()
    step50()
    t0 = time.perf_counter(); step50(); t50 = time.perf_counter() - t0
    print(thr, "threads: RN18 2x32 fwd+bwd", round(sorted(ts)[1], 3), "s; RN50 2x8", round(t50, 3), "s", flush=True)
