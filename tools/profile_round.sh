#!/bin/bash
# usage (on the GPU box, from the repository root): bash tools/profile_round.sh <tag> <workload>
# One rocprofv3 kernel-trace/stats pass and two counter passes (FETCH_SIZE, WRITE_SIZE,
# each a run of its own) over `bench.py --workload <workload>`; results under
# gpurun_out/prof_<tag>/, summarised by tools/profile_summary.py.
tag=$1
w=${2:-c2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $out/trace_bench.json 2> $out/trace.err || exit 1
echo "trace pass done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $out/fetch_bench.json 2> $out/fetch.err || exit 1
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $out/write_bench.json 2> $out/write.err || exit 1
echo "write pass done"
python3 $root/tools/profile_summary.py $out $w
