#!/bin/bash
# usage: scratch/pmc2.sh <tag> <mode> <counters...>
tag=$1; mode=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
export MODES=$mode
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/scratch/ablate.py c2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
echo "pass $tag exit $?"
