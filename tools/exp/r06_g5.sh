set -x
mkdir -p gpurun_out/r06e
(timeout 900 python -m pytest tests/test_pair_gpu.py -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" | grep -E "err|pair|passed|failed|FAILED|Error|assert" | tail -70) > gpurun_out/r06e/pair_tests.log 2>&1
tail -12 gpurun_out/r06e/pair_tests.log
(timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>gpurun_out/r06e/bench_pair.err | tail -1) > gpurun_out/r06e/bench_pair.json
(PECLR_X6_PAIR=0 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>gpurun_out/r06e/bench_six.err | tail -1) > gpurun_out/r06e/bench_six.json
cut -c1-300 gpurun_out/r06e/bench_pair.json; cut -c1-300 gpurun_out/r06e/bench_six.json; tail -5 gpurun_out/r06e/bench_pair.err
(timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -60) > gpurun_out/r06e/all_tests.log 2>&1
tail -15 gpurun_out/r06e/all_tests.log
