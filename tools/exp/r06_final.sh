mkdir -p gpurun_out/r06z
(timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5) > gpurun_out/r06z/smoke.log 2>&1; cat gpurun_out/r06z/smoke.log
(timeout 1500 python bench.py 2>gpurun_out/r06z/bench.err | tail -1) > gpurun_out/r06z/bench_fp32_n1.json
python - <<'PY'
import json
s=open('gpurun_out/r06z/bench_fp32_n1.json').read().strip().splitlines()[-1]
d=json.loads(s); print(len(s), d['ms_per_step'], d['value'], d['dtype'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], d['bf16_companion']['ms_per_step'], d['loss_delta_vs_oracle'])
print(d['fp32_gemm_check']); print(d['config']['fp32_gemm'])
PY
