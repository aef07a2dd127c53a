#!/bin/bash
# same-box A/B of the re-scheduled weight-gradient loops (gemm_x6w2_kernel, gemm_x6t2_kernel): C2 fp32, whole step as one hipGraph
cd "$(dirname "$0")/../.."
run() { echo "== $1"; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_details.json')); k=d['kernels']
print(d['ms_per_step'], d['value'], d.get('loss'), {n: (k[n]['launches'], k[n]['avg_us']) for n in k if 'wgrad' in n})"; }
for i in 1 2; do
for v in ${VARIANTS:-"PECLR_X6W2=0 PECLR_X6T2=0" "PECLR_X6W2=1 PECLR_X6T2=0" "PECLR_X6W2=1 PECLR_X6T2=1"}; do run "$v"; done
done
