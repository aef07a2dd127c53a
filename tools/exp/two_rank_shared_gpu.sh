export PECLR_DIST_BACKEND=gloo PECLR_SHARE_DEVICE=1 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611
RANK=1 LOCAL_RANK=1 python bench.py --gpus 2 --steps 10 --warmup 3 > /dev/null 2> gpurun_out/r02f_n2_rank1.err &
RANK=0 LOCAL_RANK=0 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r02f_bench_n2_shared_gpu.json 2> gpurun_out/r02f_n2_rank0.err
wait
python -c "
import json
d=json.loads(open('gpurun_out/r02f_bench_n2_shared_gpu.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['n_gpus'], d['config']['launch'][:60], d['loss_delta_vs_oracle'], d['dist']['ranks_seen'], d['dist']['backend'])"
