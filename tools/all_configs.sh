#!/bin/bash
# usage (on the GPU box): bash tools/all_configs.sh <out file>
# One bench line a workload (decode timed, re-encode leg, no CPU leg), appended to <out file>.
set -e
out=$1
: > $out
for w in c1 c2 c2v c3 c4 c5 c5g; do
    timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --steps 30 >> $out
done
