"""Copy the summaries of a tools/profile_round.sh run from gpurun_out/ into
profiles/ (tracked) and refresh profiles/traffic.json, which bench.py reads for
`roofline.traffic`.  usage: python tools/collect_profiles.py <tag> <name> [workload]"""
import glob, json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "c2"
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
stats = glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, os.path.join(dst, f"{name}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, f"{name}_summary.json"))
shutil.copy(os.path.join(src, "trace_bench.json"), os.path.join(dst, f"{name}_bench_under_rocprof.json"))
summary = json.load(open(os.path.join(src, "summary.json")))
tpath = os.path.join(dst, "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
# kernel names as bench.py knows them: no "void", no template arguments
traffic[workload] = {re.sub(r"^void ", "", re.sub(r"<.*>$", "", k)): v["total"]
                     for k, v in summary["traffic_bytes_per_launch"].items()}
traffic["_source"] = traffic.get("_source", {})
traffic["_source"][workload] = {"file": f"profiles/{name}_summary.json", "csrc_sha": summary.get("csrc_sha"),
                                "how": "rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) and WRITE_SIZE, separate passes, bytes per launch"}
json.dump(traffic, open(tpath, "w"), indent=1)
print(open(tpath).read())
