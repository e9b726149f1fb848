"""Kernel times of the inflate: one batch of a thousand small records (keys, values of words, two headers)
compressed on the host and repeated COUNT times (8192 by default).  CODEC=lz4 in the environment: an LZ4 frame
(independent blocks of 64 KiB, no checksums: what Kafka's clients write) instead of a Snappy block; CODEC=gzip: a
gzip member from zlib at its default level; CODEC=zstd: a frame from libzstd (pyarrow's) at level 3.
python tools/inflate_time.py [COUNT]"""
import json, os, struct, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import oracle
from snappy_cases import compress
from tansu_amd import columns as C, hip
from util import build_batch

count = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(5)
words = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta"]
records = []
for i in range(1000):
    value = b" ".join(words[int(x)] for x in rng.integers(0, 8, 10))
    headers = [C.Header(b"trace", bytes(rng.integers(0, 256, 8, dtype=np.uint8))), C.Header(b"kind", words[i % 8])]
    records.append(C.Record(0, 0, i, i, b"user-%06d" % int(rng.integers(0, 5000)), value, headers))
plain = build_batch(dict(base_offset=0, last_offset_delta=999), records)
lz4 = os.environ.get("CODEC") in ("lz4", "gzip", "zstd")    # (below: no count of elements)
if os.environ.get("CODEC") == "gzip":
    import gzip_cases
    data = gzip_cases.member(plain[61:])
elif os.environ.get("CODEC") == "zstd":
    import zstd_cases
    data = zstd_cases.without_size(zstd_cases.compress(plain[61:], 3))     # (as a streaming compressor writes it)
elif lz4:
    import lz4_cases
    data = lz4_cases.frame(plain[61:])
else:
    data = compress(plain[61:], window=65535)   # (as far back as a Snappy compressor's 64 KiB blocks reach)
head = bytearray(plain[:61])
struct.pack_into(">i", head, 8, 49 + len(data))
struct.pack_into(">h", head, 21, {"gzip": 1, "lz4": 3, "zstd": 4}.get(os.environ.get("CODEC"), 2))
struct.pack_into(">I", head, 17, oracle.crc32c(bytes(head[21:]) + data))
batch = bytes(head) + data
# elements of the stream
n_elem, s = 0, 0
while not lz4 and data[s] & 0x80:
    s += 1
s += 1
while not lz4 and s < len(data):
    t = data[s]
    s += 1
    if t & 3 == 0:
        ln = (t >> 2) + 1
        if ln > 60:
            ex = ln - 60
            ln = int.from_bytes(data[s:s + ex], "little") + 1
            s += ex
        s += ln
    else:
        s += {1: 1, 2: 2, 3: 4}[t & 3]
    n_elem += 1
blob = np.frombuffer(batch * count, dtype=np.uint8)
off = np.arange(count + 1, dtype=np.uint64) * len(batch)
if os.environ.get("LIB"):
    hip.LIB_PATH = os.path.abspath(os.environ["LIB"])
codec = hip.Codec(0)
d_blob, d_off = codec.to_device(blob, off)
d_out, d_out_off, d_status, totals = codec.inflate_dev(d_blob, len(blob), d_off, count)
assert int((d_status[:count] != 0).sum()) == 0
one = d_out[:len(plain)].cpu().numpy().tobytes()
assert one[:17] == plain[:17] and one[21:] == plain[21:]
codec.profile(True)
for i in range(5):
    codec.inflate_into(d_blob, len(blob), d_off, count, d_out, len(d_out) - 16, d_out_off, d_status)
torch.cuda.synchronize()
res = {k: round(ms / n, 4) for k, (ms, n) in codec.kernel_times().items()}
write = sum(res.values())   # (the call: plan, an LZ4 frame's walk into the scratch area, the walk that writes)
# the CPU restatement on one core, over a sample
import time
sample = min(count, 64)
t0 = time.perf_counter()
st, out, out_off, status = oracle.inflate_frame(blob[:sample * len(batch)], off[:sample + 1])
cpu_s = time.perf_counter() - t0
assert st == 0 and len(out) == sample * len(plain)
print(json.dumps({"cpu_one_core_gbs_out": round(len(out) / cpu_s / 1e9, 3), "batches": count, "compressed_bytes": len(blob), "inflated_bytes": int(totals.out_bytes),
                  "codec": os.environ.get("CODEC", "snappy"), "elements_a_batch": n_elem or None,
                  "bytes_an_element": round((len(plain) - 61) / n_elem, 1) if n_elem else None, "kernels_ms": res,
                  "call_ms": round(write, 3), "call_gbs_out": round(totals.out_bytes / write / 1e6, 1) if write else None}))
