#!/bin/bash
# rocprofv3 passes behind profiles/r03_* and profiles/pmc_traffic.json: kernel trace + two SQ counter passes + the two
# HBM-traffic passes (FETCH_SIZE and WRITE_SIZE need separate passes: TCC slots) (8 SQ slots per pass; counters
# only with --kernel-trace, never with the sys/hip/hsa trace domains), for (a) the labelled MFMA probe and
# (b) the eager default bench.  Run from the repo root on the MI355X box:  bash tools/profile_passes.sh <outdir>
set -u
OUT=${1:-gpurun_out/r03_prof}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
PMC_A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
PMC_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
run() {  # name, pmc ("" = trace only), command...
  local name=$1 pmc=$2; shift 2
  local d="$ROOT/$OUT/$name"
  mkdir -p "$d"
  if [ -z "$pmc" ]; then
    (cd /tmp && PECLR_LAUNCH_MANIFEST="$d/manifest.json" timeout 900 rocprofv3 --kernel-trace --stats -d "$d" -o p -- "$@" > "$d/stdout.txt" 2> "$d/stderr.txt")
  else
    (cd /tmp && PECLR_LAUNCH_MANIFEST="$d/manifest.json" timeout 900 rocprofv3 --kernel-trace --pmc $pmc -d "$d" -o p -- "$@" > "$d/stdout.txt" 2> "$d/stderr.txt")
  fi
  echo "$name rc=$? $(ls "$d" | tr '\n' ' ')"
}
BENCH="python $ROOT/bench.py --graph 0 --steps 2 --warmup 2 --no-cpu-baseline"
run probe_trace "" python "$ROOT/tools/mfma_probe.py"
run probe_pmc_a "$PMC_A" python "$ROOT/tools/mfma_probe.py"
run probe_pmc_b "$PMC_B" python "$ROOT/tools/mfma_probe.py"
run bench_trace "" $BENCH
run bench_pmc_a "$PMC_A" $BENCH
run bench_pmc_b "$PMC_B" $BENCH
run bench_pmc_fetch "FETCH_SIZE" $BENCH
run bench_pmc_write "WRITE_SIZE" $BENCH
cd "$ROOT"
python tools/pmc_traffic.py --table "$OUT/bench_pmc_fetch/p_results.db" "$OUT/bench_pmc_write/p_results.db" "$OUT/bench_pmc_fetch/manifest.json" "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.txt" 2>&1
python tools/pmc_mfma.py "$OUT/probe_trace/p_results.db" "$OUT/probe_trace/manifest.json" "$OUT/probe_mfma.json" \
  "$OUT/probe_pmc_a/p_results.db" "$OUT/probe_pmc_b/p_results.db" > "$OUT/probe_mfma.txt" 2>&1
python tools/pmc_mfma.py "$OUT/bench_trace/p_results.db" "$OUT/bench_trace/manifest.json" "$OUT/bench_mfma.json" \
  "$OUT/bench_pmc_a/p_results.db" "$OUT/bench_pmc_b/p_results.db" > "$OUT/bench_mfma.txt" 2>&1
python tools/rocpd_stats.py "$OUT/bench_trace/p_results.db" 80 > "$OUT/bench_kernel_trace_stats.txt" 2>&1
python tools/step_breakdown.py "$OUT/bench_trace/p_results.db" "$OUT/bench_trace/manifest.json" > "$OUT/step_breakdown.txt" 2>&1
# the databases are large: keep the summaries, drop the raw files beyond the 64 MiB merge limit
du -sh "$OUT"/*/ | tail -8
find "$OUT" -name "*.db" -size +20M -delete
