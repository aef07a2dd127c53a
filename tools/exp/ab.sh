#!/bin/bash
# usage: ab.sh "<bench args>" TAG1,TAG2 ENVSET1 ENVSET2 ...   (an ENVSET is "A=1,B=2" or "-" for none): alternating same-box runs
ARGS=$1; TAGS=${2//,/ }; shift 2
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for set in "$@"; do
    name=$(echo "$set" | tr ',=/' '___' | tail -c 40)
    envs=""; [ "$set" != "-" ] && envs=$(echo "$set" | tr ',' ' ')
    env $envs python bench.py $ARGS --no-cpu-baseline > gpurun_out/ab/${name}_$rep.log 2>/dev/null
    python tools/exp/ab_line.py gpurun_out/ab/${name}_$rep.log $TAGS | grep -v "^    \(\[\|/\)" | head -12
  done
done
