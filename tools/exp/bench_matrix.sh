# a few off-headline configurations through bench.py: do they run, and does the loss match the oracle
for args in "--resnet 18 --pairs 32" "--size 128 --pairs 64" "--pairs 96" "--resnet 101 --pairs 32" "--size 448 --pairs 16" "--dtype bf16" "--dtype fp16" "--checkpoint 1 --pairs 64" "--accum 2 --pairs 32" "--overlap-wgrad 1"; do
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline $args 2>/tmp/err.txt | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$args', '->', d['value'], 'img/s', d['ms_per_step'], 'ms loss', d['loss'], 'delta', d.get('loss_delta_vs_oracle'))
except Exception as e:
    print('$args', 'FAILED', e); print(open('/tmp/err.txt').read()[-800:])
"
done
