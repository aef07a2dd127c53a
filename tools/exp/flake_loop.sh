#!/bin/bash
# usage: flake_loop.sh N [ENV=VAL ...]  -- repeat the two-consumer BasicBlock test behind a few predecessors, count failures
N=$1; shift
fails=0
for i in $(seq 1 $N); do
  out=$(env "$@" timeout 300 python -m pytest tests/test_hip_parity.py tests/test_round4_gpu.py -x -q -m gpu -k "x6_tn or x6t_stride or basicblock_input" 2>&1 | grep -E "passed|failed" | tail -1)
  case "$out" in *failed*) fails=$((fails+1));; esac
done
echo "$* : $fails failures of $N"
