#!/bin/bash
# usage (on the GPU box): bash scratch/ring_knob.sh "<workloads>" "<thresholds>": decode step by ring threshold
set -e
for w in $1; do
  for t in $2; do
    NISSHI_WIDE_RING_FROM=$t timeout -k 10 120 python3 bench.py --workload $w --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', 'from $t', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels_ms'].items()})"
  done
done
