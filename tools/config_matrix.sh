#!/bin/bash
# The other BASELINE.json config shapes on ONE GPU (functional checks + per-GPU throughput), default launch mode.
# Usage: bash tools/config_matrix.sh > profiles/<round>_config_matrix.txt
for flags in "--dtype bf16 --steps 10 --warmup 3" "--resnet 152 --accum 16 --steps 2 --warmup 1" \
             "--size 448 --pairs 64 --steps 5 --warmup 2" "--size 448 --pairs 64 --dtype bf16 --steps 5 --warmup 2" \
             "--resnet 18 --pairs 32 --steps 10 --warmup 3"; do
  echo "== $flags"
  timeout 900 python bench.py --no-cpu-baseline $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'dtype', 'loss')}, d['config']['workload'], '|', d['config']['launch'])"
done
