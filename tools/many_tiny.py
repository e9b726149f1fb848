"""A million one-record batches: decode and re-encode, timings and the bit-exact check."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from tansu_amd import hip, synth
codec = hip.Codec(0)
shape = synth.Shape(key_len=4, val_len=6, n_headers=0)
n_b = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
blob, off, nr, nh = synth.make_batches(shape, np.full(n_b, 1), 1234)
d_blob, d_off = codec.to_device(blob, off)
cols = hip.DeviceColumns(n_b, nr, 4096, codec.device)
t = codec.decode_dev(d_blob, len(blob), d_off, n_b, cols)
crc = cols.t["crc_computed"][:n_b].clone()
starts = d_off[:n_b] + 17
for k in range(4):
    d_blob[starts + k] = ((crc >> (8 * (3 - k))) & 0xFF).to(torch.uint8)
codec.profile(True)
for _ in range(5):
    t = codec.decode_dev(d_blob, len(blob), d_off, n_b, cols)
assert (t.n_records, t.n_failed, t.n_exact) == (nr, 0, 0), (t.n_records, t.n_failed, t.n_exact)
out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
for _ in range(5):
    te = codec.encode_dev(cols, n_b, d_blob, cols.t["data_off"], out, len(blob), out_off)
assert te.out_bytes == len(blob) and bool((out[:len(blob)] == d_blob[:len(blob)]).all())
print(n_b, "batches", len(blob), "bytes", {k: round(ms / n, 4) for k, (ms, n) in codec.kernel_times().items()})
