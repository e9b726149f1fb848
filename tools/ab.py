"""Decode step and kernel times of one build of the library over several workloads.
python tools/ab.py <lib.so> <workload> [<workload> ...] [--enc]
One JSON line a workload: the step (wall clock over 30 enqueue-only decodes, no
event marks), the kernels (a second pass with the marks on), a checksum of the
columns so that two builds can be compared, and with --enc the re-encode step.
Builds to compare go to scratch/libs/ (not tracked)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tansu_amd import hip, shard  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
enc = "--enc" in sys.argv
hip.LIB_PATH = os.path.abspath(args[0])
hip.ABI_VERSION = None      # (older builds, which end their columns a field earlier, ignore the last one)
import torch  # noqa: E402

codec = hip.Codec(0)
for name in args[1:]:
    scale = 1.0
    if ":" in name:
        name, s = name.split(":")
        scale = float(s)
    shape, blob, off, nr, nh, nb = shard.make_shard(name, scale, 0, 1)
    d_blob, d_off = codec.to_device(blob, off)
    cols = hip.DeviceColumns(nb, nr, max(nh, 1), codec.device)
    # the CRC fields from the codec's own CRC (as bench.py does), so that a re-encode can equal the input
    codec.decode_dev(d_blob, len(blob), d_off, nb, cols)
    crc = cols.t["crc_computed"][:nb].clone()
    starts = d_off[:nb] + 17
    for k in range(4):
        d_blob[starts + k] = ((crc >> (8 * (3 - k))) & 0xFF).to(torch.uint8)
    for i in range(3):
        t = codec.decode_dev(d_blob, len(blob), d_off, nb, cols)
    assert (t.n_records, t.n_headers) == (nr, nh), (t.n_records, t.n_headers, nr, nh)
    torch.cuda.synchronize()
    steps = 30
    t0 = time.perf_counter()
    for i in range(steps):
        codec.decode_dev(d_blob, len(blob), d_off, nb, cols, sync=False)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / steps * 1e3
    codec.profile(True)
    for i in range(20):
        codec.decode_dev(d_blob, len(blob), d_off, nb, cols, sync=False)
    torch.cuda.synchronize()
    res = {k.replace("_kernel", ""): round(ms / n, 4) for k, (ms, n) in codec.kernel_times().items()}
    codec.profile(False)
    # a checksum of what the decode left: sums of the record and header columns and the CRCs
    chk = 0
    for k in ("rkey", "rval", "timestamp_delta", "hdr", "crc_computed", "rec_base", "hdr_base", "hdr_count", "status"):
        chk = (chk * 1000003 + int(cols.t[k].long().sum().item())) % (1 << 61)
    line = {"workload": name, "lib": os.path.basename(args[0]), "step_ms": round(step_ms, 4), "kernels": res,
            "failed": t.n_failed, "exact": t.n_exact, "chk": chk}
    if enc:
        out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
        out_off = torch.zeros(nb + 1, dtype=torch.int64, device=codec.device)
        te = codec.encode_dev(cols, nb, d_blob, cols.t["data_off"], out, len(blob), out_off)
        line["enc_exact"] = te.out_bytes == len(blob) and bool(torch.equal(out[:len(blob)], d_blob[:len(blob)]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            codec.encode_dev(cols, nb, d_blob, cols.t["data_off"], out, len(blob), out_off, sync=False)
        torch.cuda.synchronize()
        line["enc_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        codec.profile(True)
        for i in range(10):
            codec.encode_dev(cols, nb, d_blob, cols.t["data_off"], out, len(blob), out_off, sync=False)
        torch.cuda.synchronize()
        line["enc_kernels"] = {k.replace("_kernel", ""): round(ms / n, 4) for k, (ms, n) in codec.kernel_times().items()}
        codec.profile(False)
        del out, out_off
    print(json.dumps(line), flush=True)
    del d_blob, d_off, cols
