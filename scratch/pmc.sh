#!/bin/bash
# usage: scratch/pmc.sh <tag> <counters...>   (one rocprofv3 --pmc pass over a short bench run)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
echo "pass $tag exit $?"
