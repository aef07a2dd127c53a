set -u
OUT=gpurun_out/r03a_prof
ROOT=$(pwd)
mkdir -p "$OUT/bench_trace"
export TMPDIR=/tmp
d="$ROOT/$OUT/bench_trace"
(cd /tmp && PECLR_LAUNCH_MANIFEST="$d/manifest.json" timeout 900 rocprofv3 --kernel-trace --stats -d "$d" -o p -- python $ROOT/bench.py --graph 0 --steps 2 --warmup 2 --no-cpu-baseline > "$d/stdout.txt" 2> "$d/stderr.txt")
cd "$ROOT"
python tools/rocpd_stats.py "$OUT/bench_trace/p_results.db" 60 > "$OUT/bench_kernel_trace_stats.txt" 2>&1
python tools/step_breakdown.py "$OUT/bench_trace/p_results.db" "$OUT/bench_trace/manifest.json" > "$OUT/step_breakdown.txt" 2>&1
cat "$OUT/step_breakdown.txt"; head -45 "$OUT/bench_kernel_trace_stats.txt"
find "$OUT" -name "*.db" -size +20M -delete
