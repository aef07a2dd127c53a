"""Kernel table of a bench log (BENCH_DETAILS line), sorted by time per step: python tools/exp/kernel_table.py LOG [N]"""
import json, sys
d = None
for ln in open(sys.argv[1]):
    if ln.startswith("BENCH_DETAILS "):
        d = json.loads(ln[14:])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print(d["ms_per_step"], "ms/step", d["dtype"], d["config"].get("launch"))
rows = sorted(((v["avg_us"] * v["launches"] / d["steps"] / 1e3, k, v) for k, v in d["kernels"].items()), reverse=True)
for ms, k, v in rows[:n]:
    print(f"{ms:6.2f} ms {v['launches'] // d['steps']:4d} x {v['avg_us']:7.1f} us  {k:26s} {v['bound']:4s} frac {v['frac']:.2f}  {v['bytes'] / 1e6:8.1f} MB {v['flops'] / 1e9:8.1f} GF")
