set -x
mkdir -p gpurun_out/mdb && cp .miopen/db/* gpurun_out/mdb/
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/mdb
run() { python bench.py "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RESULT', '$*', d['value'], d['ms_per_step'])"; }
tune() { MIOPEN_FIND_ENFORCE=4 timeout 900 python bench.py "$@" --graph 0 --miopen-find 1 --steps 2 --warmup 1 > /dev/null 2> gpurun_out/tune.err; echo tune rc=$?; }
run --resnet 18 --pairs 32 --steps 20 --warmup 5
tune --resnet 18 --pairs 32
run --resnet 18 --pairs 32 --steps 20 --warmup 5
run --resnet 152 --steps 10 --warmup 3
tune --resnet 152
run --resnet 152 --steps 10 --warmup 3
run --steps 20 --warmup 5
grep -c BF16 gpurun_out/mdb/*.udb.txt; grep -c FP16 gpurun_out/mdb/*.udb.txt; grep -c FP32 gpurun_out/mdb/*.udb.txt
