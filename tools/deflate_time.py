"""Time and ratio of the deflate (CODEC=lz4, the default, gzip or zstd in the environment): the batch of
tools/inflate_time.py (a thousand small records: keys, values of words, two headers) repeated COUNT times (8192 by
default), compressed on the device; liblz4 (pyarrow's) or zlib on the same record data for the ratio.
python tools/deflate_time.py [COUNT]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
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
blob = np.frombuffer(plain * count, dtype=np.uint8)
off = np.arange(count + 1, dtype=np.uint64) * len(plain)
codec = hip.Codec(0)
d_blob, d_off = codec.to_device(blob, off)
which = os.environ.get("CODEC", "lz4")
d_codec = torch.full((count,), {"lz4": 3, "gzip": 1, "zstd": 4}[which], dtype=torch.uint8, device=codec.device)
res = {"codec": which, "batches": count, "plain_bytes": len(blob)}
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d_out, d_out_off, d_status, totals = codec.deflate_dev(d_blob, len(blob), d_off, count, d_codec)
    torch.cuda.synchronize()
    res["call_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
assert totals.n_failed == 0
res["compressed_bytes"] = int(totals.out_bytes)
res["ratio"] = round(totals.out_bytes / len(blob), 3)
res["call_gbs_in"] = round(len(blob) / res["call_ms"] / 1e6, 1)
if which == "gzip":
    import zlib
    for level in (1, 6):
        res[f"zlib_level{level}_ratio"] = round((61 + 18 + len(zlib.compress(plain[61:], level)) - 6) / len(plain), 3)
elif which == "zstd":
    import pyarrow as pa
    for level in (1, 3):
        res[f"libzstd_level{level}_ratio"] = round((61 + len(pa.Codec("zstd", compression_level=level).compress(plain[61:], asbytes=True))) / len(plain), 3)
else:
    try:
        import pyarrow as pa
        res["liblz4_ratio"] = round((61 + len(pa.compress(plain[61:], codec="lz4", asbytes=True))) / len(plain), 3)
    except Exception as e:  # noqa: BLE001
        res["liblz4_ratio"] = str(e)
# what came out inflates to what went in
back, back_off, st, t3 = codec.inflate_dev(d_out, totals.out_bytes, d_out_off, count)
got = back[:t3.out_bytes].cpu().numpy()
assert t3.out_bytes == len(blob) and (got.reshape(count, -1)[:, 21:] == blob.reshape(count, -1)[:, 21:]).all()
res["round_trip"] = True
print(json.dumps(res))
