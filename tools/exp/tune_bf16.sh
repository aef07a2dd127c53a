set -x
mkdir -p gpurun_out/mdb && cp .miopen/db/* gpurun_out/mdb/
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/mdb
python bench.py --dtype bf16 --steps 20 --warmup 5 > gpurun_out/t_bf16_before.json 2>/dev/null
date
MIOPEN_FIND_ENFORCE=4 timeout 1500 python bench.py --dtype bf16 --graph 0 --miopen-find 1 --steps 2 --warmup 1 > gpurun_out/t_bf16_tune.json 2> gpurun_out/t_bf16_tune.err
echo tune rc=$?
date
python bench.py --dtype bf16 --steps 20 --warmup 5 > gpurun_out/t_bf16_after.json 2>/dev/null
python bench.py --dtype bf16 --steps 20 --warmup 5 --miopen-find 1 > gpurun_out/t_bf16_after_find.json 2>/dev/null
for f in before after after_find; do python -c "
import json,sys
d=json.loads(open('gpurun_out/t_bf16_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
ls -la gpurun_out/mdb; grep -c BF16 gpurun_out/mdb/*.udb.txt
