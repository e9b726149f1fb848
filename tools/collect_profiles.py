"""Copy the summaries of a tools/profile_all.sh run from gpurun_out/ into profiles/
(tracked): for every workload the rocprofv3 kernel stats, the summary made from them
and from the two counter passes, and the bench line printed under the profiler, all
of one run and one build.  The build is named in profiles/<tag>_MANIFEST.json (the
hash of the kernel sources, per file copied), so that a reader can tell which files
belong together, and profiles/traffic.json, which bench.py reads for
`roofline.traffic`, is refreshed from the same summaries.
usage: python tools/collect_profiles.py <round tag> [workload ...]"""
import glob, json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
workloads = sys.argv[2:] or ["c1", "c2", "c2v", "c3", "c4", "c5", "c5g"]
dst = os.path.join(root, "profiles")
tpath = os.path.join(dst, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
mpath = os.path.join(dst, f"{tag}_MANIFEST.json")
manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
for workload in workloads:
    src = os.path.join(root, "gpurun_out", f"prof_{tag}_{workload}")
    name = f"{tag}_{workload}"
    summary = json.load(open(os.path.join(src, "summary.json")))
    # (gpurun merges a run's files into what earlier runs left: the newest is this run's)
    stats = max(glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    copies = {f"{name}_kernel_stats.csv": stats, f"{name}_summary.json": os.path.join(src, "summary.json"),
              f"{name}_bench_under_rocprof.json": os.path.join(src, "trace_bench.json")}
    for to, frm in copies.items():
        shutil.copy(frm, os.path.join(dst, to))
        manifest[to] = {"csrc_sha": summary.get("csrc_sha"), "workload": workload,
                        "command": f"bash tools/profile_round.sh {tag}_{workload} {workload}"}
    # kernel names as bench.py knows them: no "void", no template arguments
    traffic[workload] = {re.sub(r"^void ", "", re.sub(r"<.*>$", "", k)): v["total"]
                         for k, v in summary["traffic_bytes_per_launch"].items()}
    traffic.setdefault("_source", {})[workload] = {
        "file": f"profiles/{name}_summary.json", "csrc_sha": summary.get("csrc_sha"),
        "how": "rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) and WRITE_SIZE, separate passes, bytes per launch"}
json.dump(traffic, open(tpath, "w"), indent=1)
json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)
shas = {v["csrc_sha"] for v in manifest.values()}
print(f"{len(manifest)} files of {tag}, builds: {sorted(shas)}")
