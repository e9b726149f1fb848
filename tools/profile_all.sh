#!/bin/bash
# usage (on the GPU box, from the repository root): bash tools/profile_all.sh <round tag, e.g. r03>
# tools/profile_round.sh over every workload; afterwards, where the repository is kept:
# python tools/collect_profiles.py <round tag>
set -e
tag=${1:-r03}
for w in c1 c2 c2v c3 c4 c5 c5g; do
  timeout -k 10 1000 bash tools/profile_round.sh ${tag}_$w $w > gpurun_out/prof_${tag}_$w.log 2>&1 || { echo "profile $w failed"; tail -5 gpurun_out/prof_${tag}_$w.log; exit 1; }
  echo "$w done"
done
