set -x
mkdir -p gpurun_out/r06h
(timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -40) > gpurun_out/r06h/all_tests.log 2>&1
tail -12 gpurun_out/r06h/all_tests.log
(timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --steps 20 --warmup 5 2>gpurun_out/r06h/bench_bf16.err | tail -1) > gpurun_out/r06h/bench_bf16.json
python -c "
import json;d=json.loads(open('gpurun_out/r06h/bench_bf16.json').read().strip().splitlines()[-1]);print('bf16', d['ms_per_step'], d['value'], d['loss_delta_vs_oracle'])"
