set -x
mkdir -p gpurun_out/r06f
(timeout 1500 python -m pytest tests/test_pair_gpu.py tests/test_round4_gpu.py -q -m gpu -s -k "step_takes or tracks or two_consumer or basicblock_input or whole_network" 2>&1 | grep -v "amdgpu.ids" | grep -E "pair|passed|failed|FAILED|Error|assert|worst" | tail -30) > gpurun_out/r06f/tests.log 2>&1
tail -12 gpurun_out/r06f/tests.log
for tr in 128 256 0; do
  (PECLR_CONV3X3_PAIR_TILE_ROWS=$tr PECLR_BENCH_DETAILS=gpurun_out/r06f/details_tr$tr.json timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1) > gpurun_out/r06f/bench_pair_tr$tr.json
  python -c "
import json;d=json.loads(open('gpurun_out/r06f/bench_pair_tr$tr.json').read().strip().splitlines()[-1]);print('tile_rows $tr', d['ms_per_step'], d['value'])"
done
(timeout 900 python tools/exp/pair_probe.py 2>&1 | grep -v "amdgpu.ids\|Warning") > gpurun_out/r06f/pair_probe.txt
grep -A16 "weight gradients" gpurun_out/r06f/pair_probe.txt; tail -8 gpurun_out/r06f/pair_probe.txt
bash tools/profile_round.sh gpurun_out/r06f_prof fp32 > gpurun_out/r06f/profile.log 2>&1
tail -5 gpurun_out/r06f/profile.log; cat gpurun_out/r06f_prof/fp32_step_breakdown.txt | head -20
