"""Summarise one tools/profile_round.sh run: per-kernel stats of the trace pass
and HBM-side traffic per launch from the two counter passes (FETCH_SIZE is in
KiB and counts 64 B per 128-byte request on gfx950, so it is doubled)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tansu_amd.build import source_hash
out = sys.argv[1]
res = {"workload": sys.argv[2] if len(sys.argv) > 2 else "c2", "csrc_sha": source_hash(), "kernels": {},
       "traffic_bytes_per_launch": {}}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        res["kernels"][r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                    "total_ns": float(r["TotalDurationNs"]), "pct": float(r["Percentage"])}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write"):
    for f in glob.glob(out + f"/{kind}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "nisshi" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    fetch = sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [0])), 1) * 1024 * 2
    write = sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [0])), 1) * 1024
    res["traffic_bytes_per_launch"][k.replace("nisshi::", "")] = {"fetch": fetch, "write": write, "total": fetch + write}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res["traffic_bytes_per_launch"], indent=1))
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["total_ns"])[:10]:
    print(f"{k:40s} calls={v['calls']:5d} avg_us={v['avg_ns'] / 1e3:9.2f} pct={v['pct']:.1f}")
