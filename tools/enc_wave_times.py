"""When the waves of enc_write_kernel (or, with --decode, of the fused decode kernel) begin and how long they
live, by record count of the batch.  W=<workload> in the environment (c4 by default).
python tools/enc_wave_times.py <lib built with NISSHI_CXXFLAGS="-DNISSHI_DIAG_ENC_TIMES -DNISSHI_DIAG_WAVE_TIMES"> [--decode]
(the diagnostic build leaves each batch's begin and duration, 100 MHz ticks, in two header columns)"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from tansu_amd import hip, shard
hip.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
codec = hip.Codec(0)
shape, blob, off, nr, nh, nb = shard.make_shard(os.environ.get("W", "c4"), 1.0, 0, 1)
d_blob, d_off = codec.to_device(blob, off)
cols = hip.DeviceColumns(nb, nr, max(nh, 1), codec.device)
for i in range(3):
    codec.decode_dev(d_blob, len(blob), d_off, nb, cols)
count = cols.t["record_count"][:nb].cpu().numpy()
out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
out_off = torch.zeros(nb + 1, dtype=torch.int64, device=codec.device)
if "--decode" not in sys.argv:
    for i in range(3):
        codec.encode_dev(cols, nb, d_blob, cols.t["data_off"], out, len(blob), out_off)
begin = cols.t["base_sequence"][:nb].cpu().numpy().astype(np.int64) & 0xffffffff
life = cols.t["partition_leader_epoch"][:nb].cpu().numpy().astype(np.int64) & 0xffffffff
size = np.diff(np.asarray(off, dtype=np.int64))
t0 = begin.min()
begin = (begin - t0) / 100.0          # microseconds
life = life / 100.0
print("kernel: first begin 0, last end %.1f us" % (begin + life).max())
groups = sorted(set(count.tolist()))
if len(groups) > 8:                       # ragged: by powers of four
    count = 4 ** np.floor(np.log(np.maximum(count, 1)) / np.log(4)).astype(np.int64)
    groups = sorted(set(count.tolist()))
for c in groups:
    m = count == c
    print(json.dumps({"records": int(c), "batches": int(m.sum()), "bytes": int(size[m].mean()),
                      "begin_us": [round(float(begin[m].min()), 1), round(float(np.median(begin[m])), 1), round(float(begin[m].max()), 1)],
                      "life_us": [round(float(life[m].min()), 1), round(float(np.median(life[m])), 1), round(float(life[m].max()), 1)],
                      "last_end_us": round(float((begin + life)[m].max()), 1)}))
