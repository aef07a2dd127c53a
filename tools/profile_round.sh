#!/bin/bash
# The per-round rocprofv3 passes (rounds 4 and 5) of the eager bench (fp32 default and bf16): kernel trace + stats, two SQ counter passes, FETCH_SIZE /
# WRITE_SIZE passes (separate: TCC slots; counters only ever together with --kernel-trace).
#   bash tools/profile_round.sh <outdir> [fp32|bf16 ...]      (on the MI355X box, from the repo root)
set -u
OUT=${1:-gpurun_out/r05_prof}; shift || true
DTYPES=${*:-fp32 bf16}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
PMC_A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
PMC_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
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
for DT in $DTYPES; do
  BENCH="python $ROOT/bench.py --graph 0 --steps 2 --warmup 2 --no-cpu-baseline --dtype $DT"
  run ${DT}_trace "" $BENCH
  run ${DT}_pmc_a "$PMC_A" $BENCH
  run ${DT}_pmc_b "$PMC_B" $BENCH
  run ${DT}_pmc_fetch "FETCH_SIZE" $BENCH
  run ${DT}_pmc_write "WRITE_SIZE" $BENCH
  cd "$ROOT"
  python tools/pmc_traffic.py --table "$OUT/${DT}_pmc_fetch/p_results.db" "$OUT/${DT}_pmc_write/p_results.db" "$OUT/${DT}_pmc_fetch/manifest.json" "$OUT/${DT}_pmc_traffic.json" > "$OUT/${DT}_pmc_traffic.txt" 2>&1
  python tools/pmc_mfma.py "$OUT/${DT}_trace/p_results.db" "$OUT/${DT}_trace/manifest.json" "$OUT/${DT}_bench_mfma.json" \
    "$OUT/${DT}_pmc_a/p_results.db" "$OUT/${DT}_pmc_b/p_results.db" > "$OUT/${DT}_bench_mfma.txt" 2>&1
  python tools/rocpd_stats.py "$OUT/${DT}_trace/p_results.db" 80 > "$OUT/${DT}_kernel_trace_stats.txt" 2>&1
  python tools/step_breakdown.py "$OUT/${DT}_trace/p_results.db" "$OUT/${DT}_trace/manifest.json" > "$OUT/${DT}_step_breakdown.txt" 2>&1
  tail -1 "$OUT/${DT}_trace/stdout.txt" > "$OUT/${DT}_bench_under_rocprof.json"
done
du -sh "$OUT"/*/ | tail -12
# the databases are large (gpurun merges at most 64 MiB back): keep the summaries only
find "$OUT" -name "*.db" -delete
