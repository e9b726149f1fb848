import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tansu_amd import hip, synth
if os.environ.get('NISSHI_LIB'): hip.LIB_PATH = os.environ['NISSHI_LIB']
codec = hip.Codec(0)
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
shape, counts, seed = synth.config(name, float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
blob, off, nr, nh = synth.make_batches(shape, counts, seed)
n_b = len(counts)
d_blob, d_off = codec.to_device(blob, off)
cols = hip.DeviceColumns(n_b, nr, max(nh, 1), codec.device)
codec.decode_dev(d_blob, len(blob), d_off, n_b, cols)
out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
for _ in range(2): codec.encode_dev(cols, n_b, d_blob, cols.t["data_off"], out, len(blob), out_off, sync=False)
torch.cuda.synchronize()
codec.profile(True)
for _ in range(5): codec.encode_dev(cols, n_b, d_blob, cols.t["data_off"], out, len(blob), out_off, sync=False)
torch.cuda.synchronize()
kt = codec.kernel_times(); codec.profile(False)
print({k: round(v[0]/v[1], 4) for k, v in kt.items()}, "total", round(sum(v[0]/v[1] for v in kt.values()), 4))
