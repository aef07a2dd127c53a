#!/bin/bash
# repeat the tests that had two-outcome histories (two-rank rehearsal, stem encoder arms, BasicBlock) on the final build
cd /root/repo
fails=0
for i in $(seq 1 15); do
  out=$(timeout 600 python -m pytest tests/test_dist_gpu_rehearsal.py tests/test_stem_gpu.py tests/test_wgrad_resched_gpu.py -x -q -m gpu -k "rehearsal or encoder_routes or resched or two_rank or rank" 2>&1 | grep -E "passed|failed" | tail -1)
  echo "$i: $out"
  case "$out" in *failed*) fails=$((fails+1));; esac
done
echo "failures: $fails of 15"
