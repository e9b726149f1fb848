"""Decode time of large compacted batches whose records change size from one to the next.
python tools/gapped_sizes_time.py [<lib.so>] [--batches N] [--records R] [--numbered] [--short]
N batches of R records (every third gone: offset deltas 0, 1, 3, 4, 6 ...; values of 32 to 96
bytes and two headers of 0 to 32, drawn a record), checked against the CPU restatement on a
sample, then timed over 20 enqueue-only decodes: one JSON line with the step, the kernels and
how many batches one wave walked whole."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tansu_amd import hip, synth  # noqa: E402

args = [a for a in sys.argv[1:] if a.endswith(".so")]


def opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


if args:
    hip.LIB_PATH = os.path.abspath(args[0])
    hip.ABI_VERSION = None
import torch  # noqa: E402

n_batches, records = opt("--batches", 512), opt("--records", 5000)
# (--numbered: the same batches with their records numbered 0, 1, 2 ..., for comparison)
# (--short: records of 21 to 53 bytes, whose lengths take one byte: a key of 4 bytes, values of 8 to 40, no headers)
short = dict(key_len=4, val_range=(8, 40), n_headers=0) if "--short" in sys.argv else {}
blob, off, nr, nh = synth.make_var_batches(n_batches, records, synth.SEED_BASE + 9,
                                           delta=None if "--numbered" in sys.argv else synth.every_third_missing, **short)
codec = hip.Codec(0)
d_blob, d_off = codec.to_device(blob, off)
cols = hip.DeviceColumns(n_batches, nr, max(nh, 1), codec.device)
codec.decode_dev(d_blob, len(blob), d_off, n_batches, cols)
crc = cols.t["crc_computed"][:n_batches].clone()
starts = d_off[:n_batches] + 17
for k in range(4):
    d_blob[starts + k] = ((crc >> (8 * (3 - k))) & 0xFF).to(torch.uint8)
for i in range(3):
    t = codec.decode_dev(d_blob, len(blob), d_off, n_batches, cols)
assert (t.n_records, t.n_headers, t.n_failed) == (nr, nh, 0), (t.n_records, t.n_headers, t.n_failed, nr, nh)
whole = codec.last_walked_whole() if hasattr(codec, "last_walked_whole") else None
torch.cuda.synchronize()
steps = 20
t0 = time.perf_counter()
for i in range(steps):
    codec.decode_dev(d_blob, len(blob), d_off, n_batches, cols, sync=False)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / steps
codec.profile(True)
codec.decode_dev(d_blob, len(blob), d_off, n_batches, cols)
kernels = {k.replace("_kernel", ""): round(ms_k / n, 4) for k, (ms_k, n) in codec.kernel_times().items()}
codec.profile(False)
# a sample against the checker: the first batches' columns
import numpy as np  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
try:
    import oracle  # noqa: E402
    from util import assert_same  # noqa: E402
    k = min(4, n_batches)
    host = blob.copy()
    host[: int(off[k])] = d_blob[: int(off[k])].cpu().numpy()
    st, want, wr, wh = oracle.decode_columns(host[: int(off[k])], off[: k + 1])
    got = cols.to_host(nr, max(nh, 1))
    assert_same(got, want, k)
    checked = k
except ImportError:
    checked = 0
print(json.dumps({"batches": n_batches, "records_a_batch": records, "bytes": int(len(blob)), "ms_per_step": round(ms, 4),
                  "records_per_s": round(nr / ms * 1e3), "walked_whole": whole, "kernels_ms": kernels,
                  "checked_batches": checked}))
