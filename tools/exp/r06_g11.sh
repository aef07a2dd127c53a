mkdir -p gpurun_out/r06k
for seed in 3 4 5 6; do for rep in 1 2; do PECLR_STEM_TEST_SEED=$seed timeout 300 python -m pytest tests/test_stem_gpu.py -q -m gpu -s -k "encoder_routes and fp32" 2>&1 | grep -E "stem arms|AssertionError|Error|passed|failed" | tr '\n' ' '; echo " seed $seed"; done; done > gpurun_out/r06k/stem_seeds.log 2>&1
cat gpurun_out/r06k/stem_seeds.log | cut -c1-300
(timeout 900 python -m pytest tests/test_pair_gpu.py tests/test_capi_abi.py -q -m gpu 2>&1 | tail -4) > gpurun_out/r06k/pair_tests.log 2>&1
tail -3 gpurun_out/r06k/pair_tests.log
(PECLR_BENCH_DETAILS=gpurun_out/r06k/details_pair.json timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1) > gpurun_out/r06k/bench_pair.json
(PECLR_X6_PAIR=0 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1) > gpurun_out/r06k/bench_six.json
for n in pair six; do python -c "
import json;d=json.loads(open('gpurun_out/r06k/bench_$n.json').read().strip().splitlines()[-1]);print('$n', d['ms_per_step'], d['value'], d['loss_delta_vs_oracle'], d['roofline']['kernel'], d['roofline']['frac'])"; done
