"""Gather what tools/evidence.sh left under gpurun_out/ into tracked files of profiles/:
<tag>_default_bench.json, <tag>_launcher_pool.json, <tag>_operators.json, <tag>_inflate.json, <tag>_deflate.json,
<tag>_gapped_sizes.json.
usage: python tools/collect_evidence.py <round tag>"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tansu_amd.build import source_hash
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def last_json(name):
    text = open(os.path.join(root, "gpurun_out", f"{tag}_{name}.json")).read().strip()
    try:
        return json.loads(text)
    except ValueError:
        for line in reversed(text.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        raise


def put(name, value):
    value = dict(value, csrc_sha=source_hash())
    json.dump(value, open(os.path.join(root, "profiles", f"{tag}_{name}.json"), "w"), indent=1)


put("default_bench", last_json("default_bench"))
put("launcher_pool", {"note": "python bench.py --gpus 2 (two ranks on the box's one GPU), --pool 1, --pool 2 (two contexts of the one GPU)",
                      "gpus2": last_json("gpus2"), "pool1": last_json("pool1"), "pool2": last_json("pool2")})
put("operators", {"note": "HIP-event kernel times (ms) at the C2 size: tools/storage_time.py (decode, produce rows, combine in groups of 8, encode) "
                          "and tools/ops_time.py (fetch, encode of its batches, compaction)",
                  "storage_time": last_json("storage_time"), "ops_time": last_json("ops_time")})
put("inflate", {c: last_json(f"inflate_{c}") for c in ("snappy", "lz4", "gzip", "zstd")})
put("deflate", {"lz4": last_json("deflate"), "gzip": last_json("deflate_gzip"), "zstd": last_json("deflate_zstd")})
put("gapped_sizes", {"note": "tools/gapped_sizes_time.py: batches of 5000 records, every third gone, sizes drawn a record; "
                             "`numbered`: the same batches with their records numbered 0, 1, 2 ...; `short`: records of 21 to 53 bytes, no headers",
                     "batches_32": last_json("gapped_sizes_32"), "batches_32_numbered": last_json("gapped_sizes_32_numbered"),
                     "batches_512": last_json("gapped_sizes_512"), "batches_512_numbered": last_json("gapped_sizes_512_numbered"),
                     "batches_512_short": last_json("gapped_sizes_512_short"),
                     "batches_512_short_numbered": last_json("gapped_sizes_512_short_numbered")})
print("collected", tag, source_hash())
