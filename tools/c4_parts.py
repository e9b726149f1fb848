"""The mixed workload's kinds of partitions one at a time and together: decode kernel times,
and with --enc those of the re-encode.
python tools/c4_parts.py <lib.so> [--enc]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from tansu_amd import hip, synth
hip.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
codec = hip.Codec(0)
def blob_of(parts):
    blob = np.concatenate([p[0] for p in parts])
    off, base = [np.zeros(1, dtype=np.uint64)], 0
    for b, o, _, _ in parts:
        off.append(np.asarray(o[1:], dtype=np.uint64) + np.uint64(base))
        base += len(b)
    return blob, np.concatenate(off), sum(p[2] for p in parts), sum(p[3] for p in parts)
sets = {"kind%d" % k: [8 * j for j in range(128) if (8 * j) % 5 == k] for k in range(5)}
sets["all"] = [8 * j for j in range(128)]
sets["kinds12"] = [8 * j for j in range(128) if (8 * j) % 5 in (1, 2)]
for k in (0, 3, 4):
    sets["without%d" % k] = [8 * j for j in range(128) if (8 * j) % 5 != k]
if os.environ.get("SETS"):
    sets = {k: v for k, v in sets.items() if k in os.environ["SETS"].split(",")}
for name, only in sets.items():
    parts = synth.mixed_partitions(1024, 64, only=only)
    blob, off, nr, nh = blob_of(parts)
    nb = len(off) - 1
    d_blob, d_off = codec.to_device(blob, off)
    cols = hip.DeviceColumns(nb, nr, max(nh, 1), codec.device)
    for i in range(3):
        t = codec.decode_dev(d_blob, len(blob), d_off, nb, cols)
    assert (t.n_records, t.n_headers, t.n_failed) == (nr, nh, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        codec.decode_dev(d_blob, len(blob), d_off, nb, cols, sync=False)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 30 * 1e3
    codec.profile(True)
    for i in range(20):
        codec.decode_dev(d_blob, len(blob), d_off, nb, cols, sync=False)
    torch.cuda.synchronize()
    res = {k.replace("_kernel", ""): round(ms / n, 4) for k, (ms, n) in codec.kernel_times().items()}
    codec.profile(False)
    enc = None
    if "--enc" in sys.argv:
        out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
        out_off = torch.zeros(nb + 1, dtype=torch.int64, device=codec.device)
        codec.profile(True)
        for i in range(10):
            codec.encode_dev(cols, nb, d_blob, cols.t["data_off"], out, len(blob), out_off, sync=False)
        torch.cuda.synchronize()
        enc = {k.replace("_kernel", ""): round(ms / n, 4) for k, (ms, n) in codec.kernel_times().items()}
        codec.profile(False)
        del out, out_off
    print(json.dumps({"set": name, "enc": enc, "batches": nb, "MB": round(len(blob) / 1e6, 1), "step_ms": round(step, 4), "kernels": res}), flush=True)
    del d_blob, d_off, cols
