# MIOpen kernel-config search (MIOPEN_FIND_ENFORCE=4) for the bench shapes that are not in the in-tree perf-db yet;
# the grown databases come back under gpurun_out/mdb (copy them into .miopen/db).
set -x
mkdir -p gpurun_out/mdb && cp .miopen/db/* gpurun_out/mdb/
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/mdb
run() { python bench.py "$@" --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RESULT', '$*', d['value'], d['ms_per_step'])"; }
tune() { MIOPEN_FIND_ENFORCE=4 timeout 1200 python bench.py "$@" --graph 0 --miopen-find 1 --steps 2 --warmup 1 > /dev/null 2> gpurun_out/tune.err; echo tune rc=$?; }
run --dtype fp16
tune --dtype fp16
run --dtype fp16
run --dtype bf16 --pairs 64 --size 448
tune --dtype bf16 --pairs 64 --size 448
run --dtype bf16 --pairs 64 --size 448
run --pairs 64 --size 448
tune --pairs 64 --size 448
run --pairs 64 --size 448
grep -c BF16 gpurun_out/mdb/*.udb.txt; grep -c FP16 gpurun_out/mdb/*.udb.txt; grep -c FP32 gpurun_out/mdb/*.udb.txt
