#!/bin/bash
# MIOpen kernel-config search for one bench configuration, growing the in-tree databases (.miopen/db/*.txt).
#
# MIOpen's asm implicit-GEMM solvers pick their tile / split-K configuration by heuristic unless the perf-db
# holds a searched entry for the convolution.  For fp32 ResNet-50 @224 the heuristic is already the fastest
# choice; for the 16-bit, 448x448 and ResNet-18 shapes the searched configurations are 4-13 % faster per step
# (DESIGN.md section 5).  The search needs the GPU: run this on the MI355X box from the repo root, e.g.
#     bash tools/miopen_search.sh --dtype bf16
#     bash tools/miopen_search.sh --dtype fp16 --pairs 64 --size 448
# It prints the step time before / after, and leaves the grown databases in .miopen/db (tracked in git; under
# gpurun copy them back through gpurun_out/: MIOPEN_DB_OUT=gpurun_out/mdb).
set -u
DB=${MIOPEN_DB_OUT:-.miopen/db}
mkdir -p "$DB"
[ "$DB" != ".miopen/db" ] && cp .miopen/db/*.txt "$DB"/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$PWD/$DB
line() { python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$*', d['value'], 'images/s', d['ms_per_step'], 'ms')"; }
line "$@"
MIOPEN_FIND_ENFORCE=4 timeout 1500 python bench.py "$@" --graph 0 --miopen-find 1 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/miopen_search.err
echo "search rc=$? ($(grep -c . "$DB"/*.udb.txt | tr '\n' ' ') perf-db lines)"
line "$@"
