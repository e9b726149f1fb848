import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from tansu_amd import hip, synth
import oracle
codec = hip.Codec(0)
for blob, off, nr, nh in synth.mixed_partitions(10, 3):
    d_blob, d_off = codec.to_device(blob, off)
    n = len(off) - 1
    cols = hip.DeviceColumns(n, nr, max(nh, 1) + 4096 * n, torch.device("cuda", 0))
    codec.decode_dev(d_blob, len(blob), d_off, n, cols)
    got = cols.t["crc_computed"][:n].cpu().numpy()
    for b in range(n):
        lo, hi = int(off[b]) + 21, int(off[b + 1])
        want = oracle.crc32c(bytes(blob[lo:hi]))
        flag = "" if want == int(got[b]) else "  <-- MISMATCH"
        print(b, "lo", lo, "len", hi - lo, "lo%16", lo % 16, "rows", ((hi - (lo & ~15)) + 1023) // 1024, hex(want), hex(int(got[b])), flag)
