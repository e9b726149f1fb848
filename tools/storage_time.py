"""Per-kernel times of the storage operators at the C2 size."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from tansu_amd import hip, synth

codec = hip.Codec(0)
name = os.environ.get("CFG", "c2")
shape, counts, seed = synth.config(name, float(os.environ.get("SCALE", "1.0")))
blob, off, nr, nh = synth.make_batches(shape, counts, seed)
n_b = len(counts)
d_blob, d_off = codec.to_device(blob, off)
cols = hip.DeviceColumns(n_b, nr, max(nh, 1) + 4096, codec.device)
t = codec.decode_dev(d_blob, len(blob), d_off, n_b, cols)
crc = cols.t["crc_computed"].cpu().numpy().view(np.uint32)[:n_b]
synth.patch_crc(blob, off, crc)
d_blob, d_off = codec.to_device(blob, off)
high = torch.arange(n_b, device=codec.device, dtype=torch.int64) * 1000
G = int(os.environ.get("GROUP", "8"))
groups = torch.from_numpy(np.arange(0, n_b + 1, G).astype(np.int32)).to(codec.device)
n_g = n_b // G
out = torch.zeros(len(blob) + 2 * nr + 64, dtype=torch.uint8, device=codec.device)
out_off = torch.zeros(n_g + 1, dtype=torch.int64, device=codec.device)
codec.profile(True)
reps = 10
for i in range(reps):
    codec.decode_dev(d_blob, len(blob), d_off, n_b, cols, sync=False)
    rows = codec.produce_rows_dev(cols, n_b, nr, high, True)
    sink = codec.combine_dev(cols, n_b, nr, groups, n_g)
    codec.encode_dev(sink, n_g, d_blob, sink.t["data_off"], out, len(out) - 16, out_off, sync=False)
torch.cuda.synchronize()
res = {k: round(ms / n, 4) for k, (ms, n) in codec.kernel_times().items()}
res["records"] = nr; res["headers"] = nh; res["batches"] = n_b; res["blob"] = len(blob)
print(json.dumps(res, indent=1))
