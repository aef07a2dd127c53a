#!/bin/bash
# kernel traces of the C5-shaped step (ResNet-50, 2 x 64 views @448) in fp32 and bf16: where the time goes, what is left on MIOpen
OUT=gpurun_out/r06_c5; ROOT=$(pwd); mkdir -p $OUT; export TMPDIR=/tmp
for DT in fp32 bf16; do
  d="$ROOT/$OUT/${DT}_trace"; mkdir -p "$d"
  (cd /tmp && PECLR_LAUNCH_MANIFEST="$d/manifest.json" timeout 900 rocprofv3 --kernel-trace --stats -d "$d" -o p -- python $ROOT/bench.py --graph 0 --steps 2 --warmup 2 --no-cpu-baseline --dtype $DT --size 448 --pairs 64 > "$d/stdout.txt" 2> "$d/stderr.txt")
  python tools/step_breakdown.py "$d/p_results.db" "$d/manifest.json" > "$OUT/${DT}_step_breakdown.txt" 2>&1
  python tools/rocpd_stats.py "$d/p_results.db" 40 > "$OUT/${DT}_kernel_trace_stats.txt" 2>&1
done
find $OUT -name "*.db" -delete
head -20 $OUT/fp32_step_breakdown.txt; head -20 $OUT/bf16_step_breakdown.txt
