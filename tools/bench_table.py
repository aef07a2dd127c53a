#!/usr/bin/env python
"""Per-tag table of a bench.py run: python tools/bench_table.py <log with the BENCH_DETAILS line> [<second log> ...]"""
import json
import sys


def table(path):
    line = [ln for ln in open(path) if ln.startswith("BENCH_DETAILS ")][-1]
    d = json.loads(line[len("BENCH_DETAILS "):])
    steps = d["steps"] * d["config"].get("accumulate_grad_batches", 1)
    rows = []
    for tag, k in d["kernels"].items():
        rows.append((k["launches"] * k["avg_us"] / steps / 1e3, tag, k["launches"] // steps, k["avg_us"], k["bound"], k["frac"], k.get("kernel")))
    rows.sort(reverse=True)
    print(f"# {path}: {d['ms_per_step']} ms/step, {d['value']} images/s, hand-written {d['hand_written_us_per_step'] / 1e3:.2f} ms/step")
    for ms, tag, n, us, bound, frac, kern in rows:
        print(f"{ms:8.3f} ms  {n:4d} x {us:9.1f} us  {bound:4s} {frac:7.4f}  {tag:32s} {kern if isinstance(kern, str) else ''}")


for p in sys.argv[1:]:
    table(p)
