#!/bin/bash
# usage (on the GPU box, from the repository root): bash tools/inst_mix.sh <workload>
# One rocprofv3 counter pass (instruction counts by kind, wave cycles) over a short
# bench run; prints the averages a launch of every nisshi kernel.
w=${1:-c2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/mix_$w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $out/a -- python3 $root/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/a.json 2> $out/a.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $out/b -- python3 $root/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/b.json 2> $out/b.err || exit 1
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "nisshi" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:24s} {sum(v) / len(v):16.0f}")
PY
