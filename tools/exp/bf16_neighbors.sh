set -u
ROOT=$(pwd); export TMPDIR=/tmp
d=/tmp/nb; mkdir -p $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $d -o p -- python $ROOT/bench.py --dtype bf16 --graph 0 --steps 2 --warmup 2 --no-cpu-baseline > $d/stdout.txt 2> $d/stderr.txt)
for pat in copyBuffer fillBuffer FillFunctor bfloat16_copy bfloat16tofloat32 SubTensorOpWithScalar SubTensorOpWithCast; do echo "== $pat"; python tools/exp/neighbors.py $d/p_results.db $pat; done
