"""Decode kernel times of a diagnostic build whose results may be wrong:
python scratch/ab_noassert.py <lib> <workload>."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from tansu_amd import hip, shard
hip.LIB_PATH = os.path.abspath(sys.argv[1])
codec = hip.Codec(0)
shape, blob, off, n_records, n_headers, n_batches = shard.make_shard(sys.argv[2], 1.0, 0, 1)
d_blob, d_off = codec.to_device(blob, off)
cols = hip.DeviceColumns(n_batches, n_records, max(n_headers, 1), torch.device("cuda", 0))
for _ in range(3):
    codec.decode_dev(d_blob, len(blob), d_off, n_batches, cols)
codec.profile(True)
for _ in range(20):
    codec.decode_dev(d_blob, len(blob), d_off, n_batches, cols, sync=False)
torch.cuda.synchronize()
print(sys.argv[1], {k: round(v[0] / v[1], 4) for k, v in codec.kernel_times().items()})
