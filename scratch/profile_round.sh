#!/bin/bash
# usage (on the GPU box): bash scratch/profile_round.sh <tag>
# One rocprofv3 kernel-trace/stats pass and two counter passes (FETCH_SIZE, WRITE_SIZE)
# over the default bench command; results under gpurun_out/prof_<tag>/.
set -e
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/trace_bench.json 2> $out/trace.err
echo "trace pass done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/fetch_bench.json 2> $out/fetch.err
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/write_bench.json 2> $out/write.err
echo "write pass done"
python3 $GRAFT_REPO_ROOT/scratch/profile_summary.py $out
