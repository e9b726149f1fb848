import os, sys, time
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
capx = int(os.environ.get('CAPX', '1'))
cols = hip.DeviceColumns(n_b, nr * capx, max(nh, 1) * capx, codec.device)
for mode in os.environ.get("MODES", "0,1,2,3").split(","):
    os.environ["NISSHI_DBG"] = mode
    for _ in range(2): codec.decode_dev(d_blob, len(blob), d_off, n_b, cols, sync=False)
    torch.cuda.synchronize()
    codec.profile(True)
    for _ in range(5): codec.decode_dev(d_blob, len(blob), d_off, n_b, cols, sync=False)
    torch.cuda.synchronize()
    kt = codec.kernel_times(); codec.profile(False)
    print("mode", mode, {k: round(v[0]/v[1], 4) for k, v in kt.items() if v[0]/v[1] > 0.01})
