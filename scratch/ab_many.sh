#!/bin/bash
# usage (on the GPU box): bash scratch/ab_many.sh "<libs>" "<workloads>": decode step and fused kernel time of each pair
set -e
for w in $2; do
  for l in $1; do
    timeout -k 10 120 python3 scratch/ab.py $l $w 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', '$l', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels_ms'].items()}, 'enc', round(d['reencode']['ms_per_step'],4))"
  done
done
