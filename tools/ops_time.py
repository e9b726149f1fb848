"""Per-kernel times of the fetch seam and of log compaction at the C2 size.
KEYS=k in the environment gives the records k distinct keys (record r gets key r mod k) instead of one each."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from tansu_amd import hip, synth

if os.environ.get("LIB"):
    hip.LIB_PATH = os.path.abspath(os.environ["LIB"])
codec = hip.Codec(0)
dev = codec.device
shape, counts, seed = synth.config(os.environ.get("CFG", "c2"), float(os.environ.get("SCALE", "1.0")))
blob, off, nr, nh = synth.make_batches(shape, counts, seed)
n_b = len(counts)
d_blob, d_off = codec.to_device(blob, off)
cols = hip.DeviceColumns(n_b, nr, max(nh, 1) + 4096, dev)
t = codec.decode_dev(d_blob, len(blob), d_off, n_b, cols)
high = torch.arange(n_b, device=dev, dtype=torch.int64) * 1000
rows = codec.produce_rows_dev(cols, n_b, nr, high, True)
torch.cuda.synchronize()
# fetch rows from the produce rows: spans as offsets into the blob, headers in record order
rkey = cols.t["rkey"][:nr].view(torch.int32).reshape(nr, 4)
rval = cols.t["rval"][:nr].view(torch.int32).reshape(nr, 4)
batch = rows.t["batch"][:nr].long()
base = cols.t["data_off"][:n_b][batch]
hc = rval[:, 3].long()
hb = torch.zeros(nr + 1, dtype=torch.int64, device=dev)
hb[1:] = torch.cumsum(hc, 0)
n_hdr = int(hb[-1])
src = torch.repeat_interleave(rval[:, 2].long() & 0xffffffff, hc) + (torch.arange(n_hdr, device=dev) - torch.repeat_interleave(hb[:-1], hc))
hdr = cols.t["hdr"].view(torch.int32).reshape(-1, 4)[src].clone()
hbase = torch.repeat_interleave(base, hc).int()
hdr[:, 0] += hbase
hdr[:, 2] += hbase
keep = dict(offset=rows.t["offset"][:nr].clone(), attributes=torch.zeros(nr, dtype=torch.int16, device=dev),
            timestamp=rows.t["timestamp"][:nr].clone(), producer_id=torch.full((nr,), -1, dtype=torch.int64, device=dev),
            producer_epoch=torch.full((nr,), -1, dtype=torch.int16, device=dev),
            completed=torch.ones(nr, dtype=torch.uint8, device=dev),
            key_off=(rkey[:, 2].long() + base).int(), key_len=rkey[:, 3].contiguous(),
            val_off=(rval[:, 0].long() + base).int(), val_len=rval[:, 1].contiguous(),
            hdr_base=hb.int(), hdr=hdr.contiguous())
# a new batch every 1000 rows, as the producer id changes
keep["producer_id"] = (torch.arange(nr, device=dev) // 1000).long()
c = hip.CFetchRows()
for name, _, _ in hip.FETCH_ARRAYS:
    setattr(c, name, keep[name].data_ptr())
n_keys = int(os.environ.get("KEYS", "0"))
if n_keys:
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    table = torch.randint(0, 256, (n_keys, 16), dtype=torch.uint8, device=dev, generator=gen)
    at = keep["key_off"].long()
    for j in range(16):
        d_blob[at + j] = table[torch.arange(nr, device=dev) % n_keys, j]
    del table, at
out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=dev)
out_off = torch.zeros(n_b + 2, dtype=torch.int64, device=dev)
codec.profile(True)
reps = 10
for i in range(reps):
    fcols, nb = codec.fetch_batches_dev(c, keep["hdr"], nr, 1_700_000_000_000, n_b + 1)
    codec.encode_dev(fcols, nb, d_blob, fcols.t["data_off"], out, len(out) - 16, out_off, sync=False)
    ccols, removed, action, tot = codec.compact_dev(d_blob, cols, n_b, nr)
torch.cuda.synchronize()
res = {k: round(ms / n, 4) for k, (ms, n) in codec.kernel_times().items()}
res.update(records=nr, headers=nh, batches=n_b, blob=len(blob), fetch_batches=nb, kept=int(tot.n_records), keys=n_keys)
# the fetched batches carry the records of the decoded ones
back = hip.DeviceColumns(nb, nr, max(nh, 1) + 4096, dev)
t2 = codec.decode_dev(out, int(out_off[nb]), out_off, nb, back)
res["fetch_roundtrip_records"] = int(t2.n_records)
print(json.dumps(res, indent=1))
