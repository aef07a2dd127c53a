"""Print ms/step and a few kernel tags from bench logs: python tools/exp/ab_line.py LOG [TAG ...]"""
import json, sys
log, tags = sys.argv[1], sys.argv[2:]
det = None
for ln in open(log):
    if ln.startswith("BENCH_DETAILS "):
        det = json.loads(ln[len("BENCH_DETAILS "):])
    elif not ln.startswith("{"):
        print("   ", ln.rstrip()[:300])
if det is None:
    sys.exit("no BENCH_DETAILS line in " + log)
print(log, det["ms_per_step"], "ms/step", det["value"], "| launch:", det["config"].get("launch"), "| latency-bound:", det.get("latency_bound_launches_per_step"))
for t in tags:
    k = det["kernels"].get(t)
    if k:
        print(f"    {t:24s} {k['launches'] // det['steps']:4d} x {k['avg_us']:8.1f} us")
