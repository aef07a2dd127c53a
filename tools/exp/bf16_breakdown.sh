# kernel trace + per-category breakdown of the eager bf16 / fp16 step (final build, searched MIOpen configurations)
set -u
ROOT=$(pwd); export TMPDIR=/tmp
for dt in bf16 fp16; do
  d="$ROOT/gpurun_out/r02x_$dt"; mkdir -p "$d"
  (cd /tmp && PECLR_LAUNCH_MANIFEST="$d/manifest.json" timeout 600 rocprofv3 --kernel-trace --stats -d "$d" -o p -- python $ROOT/bench.py --dtype $dt --graph 0 --steps 2 --warmup 2 --no-cpu-baseline > "$d/stdout.txt" 2> "$d/stderr.txt")
  python tools/rocpd_stats.py "$d/p_results.db" 60 > "$d/kernel_trace_stats.txt" 2>&1
  python tools/step_breakdown.py "$d/p_results.db" "$d/manifest.json" > "$d/step_breakdown.txt" 2>&1
  find "$d" -name "*.db" -size +20M -delete
  cat "$d/step_breakdown.txt"
done
