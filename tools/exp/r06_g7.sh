set -x
mkdir -p gpurun_out/r06g
(timeout 900 python -m pytest tests/test_pair_gpu.py -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -8) > gpurun_out/r06g/pair_tests.log 2>&1
tail -4 gpurun_out/r06g/pair_tests.log
(PECLR_BENCH_DETAILS=gpurun_out/r06g/details_pair.json timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>gpurun_out/r06g/bench_pair.err | tail -1) > gpurun_out/r06g/bench_pair.json
(PECLR_X6_PAIR=0 PECLR_BENCH_DETAILS=gpurun_out/r06g/details_six.json timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>gpurun_out/r06g/bench_six.err | tail -1) > gpurun_out/r06g/bench_six.json
for n in pair six; do python -c "
import json;d=json.loads(open('gpurun_out/r06g/bench_$n.json').read().strip().splitlines()[-1]);print('$n', d['ms_per_step'], d['value'], d['loss_delta_vs_oracle'])"; done
bash tools/profile_round.sh gpurun_out/r06g_prof fp32 > gpurun_out/r06g/profile.log 2>&1
head -16 gpurun_out/r06g_prof/fp32_step_breakdown.txt; grep -E "x6_|bn2d_apply |bn2d_bwd_apply " gpurun_out/r06g_prof/fp32_bench_mfma.txt | cut -c1-120
