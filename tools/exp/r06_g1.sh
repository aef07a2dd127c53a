set -x
mkdir -p gpurun_out/r06a
(timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_stem_gpu.py tests/test_bn_shortcut_in_add_gpu.py tests/test_graph_lifetime_gpu.py tests/test_dist_gpu_rehearsal.py -q -m gpu -x -k "golden or align_single or stem_non_finite or stem_fp32 or shortcut or other_reader or graph or two_ranks" 2>&1 | tail -15) > gpurun_out/r06a/tests.log 2>&1
(timeout 1200 python tools/exp/rehearsal_noise.py 10 2>&1 | grep -v Warning | tail -30) > gpurun_out/r06a/noise.log 2>&1
(timeout 900 python bench.py 2>gpurun_out/r06a/bench.err | tail -1) > gpurun_out/r06a/bench_fp32_n1.json
tail -3 gpurun_out/r06a/tests.log; tail -4 gpurun_out/r06a/noise.log; cut -c1-400 gpurun_out/r06a/bench_fp32_n1.json
