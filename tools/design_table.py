"""The table of measured workloads as DESIGN.md shows it, from profiles/<tag>_all_configs.jsonl (bench lines,
tools/all_configs.sh) and profiles/traffic.json (counter passes).  usage: python tools/design_table.py r03"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
traffic = json.load(open(os.path.join(root, "profiles", "traffic.json")))
keys = ["c1", "c2", "c2v", "c3", "c4", "c5", "c5g"]
print("| Workload | decode step | records/s | dominant kernel: ms, fraction of 8 TB/s | HBM traffic / algorithmic | re-encode (bit-exact) |")
print("|---|---|---|---|---|---|")
for key, line in zip(keys, open(os.path.join(root, "profiles", f"{tag}_all_configs.jsonl"))):
    j = json.loads(line)
    r = j["roofline"]
    t = traffic.get(key, {}).get(r["kernel"])
    alg = r["kernel_algorithmic_bytes"]
    tr = f"{t / 1e9:.3f} GB / {alg / 1e9:.3f} GB = {t / alg:.2f}" if t else "-"
    re = j["reencode"]
    print(f"| {j['config']['workload'].split(':')[0]} ({j['config']['input_bytes_per_gpu'] / 1e9:.2f} GB) | {j['ms_per_step']:.3f} ms | "
          f"{j['value'] / 1e9:.1f} G | `{r['kernel'].replace('_kernel', '')}` {r['kernel_ms']:.3f}, {r['frac']:.2f} | {tr} | "
          f"{re['ms_per_step']:.3f} ms ({'yes' if re['bit_exact'] else 'NO'}) |")
