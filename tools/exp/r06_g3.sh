set -x
mkdir -p gpurun_out/r06c
(timeout 1200 python -m pytest tests/test_pair_gpu.py -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" | tail -60) > gpurun_out/r06c/pair_tests.log 2>&1
(timeout 1200 python tools/exp/pair_probe.py 2>&1 | grep -v "amdgpu.ids\|Warning") > gpurun_out/r06c/pair_probe.txt
(timeout 900 python -m pytest tests/test_graph_lifetime_gpu.py tests/test_hip_parity.py -q -m gpu -x -k "graph or x6 or conv3x3" 2>&1 | tail -5) > gpurun_out/r06c/tests.log 2>&1
tail -25 gpurun_out/r06c/pair_tests.log; cat gpurun_out/r06c/pair_probe.txt | tail -70; tail -3 gpurun_out/r06c/tests.log
