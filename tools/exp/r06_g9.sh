set -x
mkdir -p gpurun_out/r06i
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_stem_gpu.py -q -m gpu -k "encoder_routes and fp32" 2>&1 | grep -E "AssertionError: \(|passed|failed" | tail -2; done > gpurun_out/r06i/stem_flake.log 2>&1
cat gpurun_out/r06i/stem_flake.log
for i in 1 2 3; do PECLR_X6_PAIR=0 timeout 300 python -m pytest tests/test_stem_gpu.py -q -m gpu -k "encoder_routes and fp32" 2>&1 | grep -E "AssertionError: \(|passed|failed" | tail -2; done > gpurun_out/r06i/stem_flake_six.log 2>&1
cat gpurun_out/r06i/stem_flake_six.log
(timeout 900 python tools/exp/pair_probe.py --real-only 2>&1 | grep -v "amdgpu.ids\|Warning") > gpurun_out/r06i/pair_probe_real.txt
grep -E "wgrad3|worst|wgrad " gpurun_out/r06i/pair_probe_real.txt | tail -40
