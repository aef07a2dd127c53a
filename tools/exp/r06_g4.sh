set -x
mkdir -p gpurun_out/r06d
(timeout 900 python -m pytest tests/test_pair_gpu.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids" | tail -40) > gpurun_out/r06d/pair_tests.log 2>&1
tail -5 gpurun_out/r06d/pair_tests.log
(timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>gpurun_out/r06d/bench_pair.err | tail -1) > gpurun_out/r06d/bench_pair.json
(PECLR_X6_PAIR=0 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>gpurun_out/r06d/bench_six.err | tail -1) > gpurun_out/r06d/bench_six.json
cut -c1-300 gpurun_out/r06d/bench_pair.json; cut -c1-300 gpurun_out/r06d/bench_six.json; tail -5 gpurun_out/r06d/bench_pair.err
(timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "amdgpu.ids" | tail -40) > gpurun_out/r06d/all_tests.log 2>&1
tail -12 gpurun_out/r06d/all_tests.log
