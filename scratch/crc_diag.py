import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, oracle
from tansu_amd import hip
codec = hip.Codec(0)
rng = np.random.default_rng(7)
buf = rng.integers(0, 256, size=300_000, dtype=np.uint8)
t = torch.from_numpy(buf).to(codec.device)
bad = []
for n in [1,4,16,1000,1024,1025,2047,2048,2049,3000,3072,3073,4096,5000,10000,65536,65537,131072,200001]:
    for start in (0,1,7,15,16,21,1003):
        ok = codec.crc32c_dev(t, n, offset=start) == oracle.crc32c(buf[start:start+n].tobytes())
        rows = ((start % 16) + n + 1023)//1024
        print(n, start, rows, "ok" if ok else "BAD")
z = torch.zeros(5000, dtype=torch.uint8, device=codec.device)
for n in (1024, 2048, 2049, 3000, 5000):
    print("zeros", n, codec.crc32c_dev(z, n) == oracle.crc32c(bytes(n)))
one = np.zeros(5000, dtype=np.uint8); 
for pos in (0, 5, 1023, 1024, 2047, 2048, 3000):
    one[:] = 0; one[pos] = 1
    tt = torch.from_numpy(one).to(codec.device)
    print("onehot", pos, [codec.crc32c_dev(tt, n) == oracle.crc32c(one[:n].tobytes()) for n in (3072, 3073, 4096, 5000)])
