#!/bin/bash
# usage: scratch/pmc3.sh <tag> <lib or -> <counters...>
tag=$1; lib=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
export MODES=0
if [ "$lib" != "-" ]; then export NISSHI_LIB=$GRAFT_REPO_ROOT/$lib; fi
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/scratch/ablate.py c2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
echo "pass $tag exit $?"; grep mode $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log
