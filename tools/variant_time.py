"""Decode kernel times of a diagnostic build of the library.
python tools/variant_time.py <lib.so> <workload> [<workload> ...]
Builds go to scratch/libs/ (not tracked): hipcc with the flags of tansu_amd/build.py plus -D switches."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tansu_amd import hip, shard
hip.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
codec = hip.Codec(0)
for name in sys.argv[2:]:
    shape, blob, off, nr, nh, nb = shard.make_shard(name, 1.0, 0, 1)
    d_blob, d_off = codec.to_device(blob, off)
    cols = hip.DeviceColumns(nb, nr, max(nh, 1), codec.device)
    for i in range(3):
        codec.decode_dev(d_blob, len(blob), d_off, nb, cols)
    codec.profile(True)
    for i in range(20):
        codec.decode_dev(d_blob, len(blob), d_off, nb, cols, sync=False)
    torch.cuda.synchronize()
    res = {k: round(ms / n, 4) for k, (ms, n) in codec.kernel_times().items()}
    codec.profile(False)
    print(name, os.path.basename(sys.argv[1]), json.dumps(res), flush=True)
    del d_blob, d_off, cols
