set -x
mkdir -p gpurun_out/r06j
for seed in 3 4 5 6 7; do for rep in 1 2; do PECLR_STEM_TEST_SEED=$seed timeout 300 python -m pytest tests/test_stem_gpu.py -q -m gpu -s -k "encoder_routes" 2>&1 | grep -E "stem arms|bf16 vs fp32|AssertionError|passed|failed" | tr '\n' ' '; echo " seed $seed"; done; done > gpurun_out/r06j/stem_seeds.log 2>&1
cat gpurun_out/r06j/stem_seeds.log
(timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --steps 20 --warmup 5 2>/dev/null | tail -1) > gpurun_out/r06j/bench_bf16.json
python -c "
import json;d=json.loads(open('gpurun_out/r06j/bench_bf16.json').read().strip().splitlines()[-1]);print('bf16', d['ms_per_step'], d['value'])"
