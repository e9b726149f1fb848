"""Phase counters of the decode walks: build with NISSHI_CXXFLAGS=-DNISSHI_PHASES into scratch/libs/phases.so first (scratch/ is not tracked).
python tools/phases.py <workload> [scale]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tansu_amd import hip, shard
hip.LIB_PATH = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "scratch", "libs", "phases.so"))
hip.ABI_VERSION = None
import torch
codec = hip.Codec(0)
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
shape, blob, off, nr, nh, nb = shard.make_shard(sys.argv[1], scale, 0, 1)
d_blob, d_off = codec.to_device(blob, off)
cols = hip.DeviceColumns(nb, nr, max(nh, 1), codec.device)
t = codec.decode_dev(d_blob, len(blob), d_off, nb, cols)
print(t.n_records, t.n_headers, t.n_failed, t.n_exact)
