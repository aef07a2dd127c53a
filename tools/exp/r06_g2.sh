set -x
mkdir -p gpurun_out/r06b
(timeout 1500 python -m pytest tests/test_stem_gpu.py tests/test_bn_shortcut_in_add_gpu.py tests/test_graph_lifetime_gpu.py tests/test_dist_gpu_rehearsal.py -q -m gpu -k "stem_non_finite or stem_fp32 or stem_weight or shortcut or other_reader or graph or two_ranks" 2>&1 | tail -15) > gpurun_out/r06b/tests.log 2>&1
(timeout 600 python tools/exp/rehearsal_noise.py 5 frozen 2>&1 | grep -v "amdgpu.ids\|Warning") > gpurun_out/r06b/noise_frozen.log 2>&1
(timeout 900 tools/exp/x6p_ablate 2>&1) > gpurun_out/r06b/x6p_ablate.txt
tail -3 gpurun_out/r06b/tests.log; tail -2 gpurun_out/r06b/noise_frozen.log; tail -5 gpurun_out/r06b/x6p_ablate.txt
