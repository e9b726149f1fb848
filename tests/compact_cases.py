"""Batches for the log compaction tests and a Python model of the reference
(nisshi-sans-io/src/record/inflated.rs:213-275, nisshi-storage/src/slate/
storage.rs:283-312), independent of the C restatement."""
import numpy as np

from tansu_amd import columns as C
import oracle
from util import body_size, build_batch


def keyed(od, key, value):
    # inflated.rs:657-662
    return C.Record(0, 0, 0, od, key, value)


def make_partition(n_batches, per_batch, n_keys, seed=0, keyless=0.1, control_every=0, dup_delta=False,
                   headers=False, long_keys=False, stored_length_off=False):
    """Deflated batches of one partition, oldest first: keys drawn from n_keys values."""
    rng = np.random.default_rng(seed)
    out, base = [], 0
    for b in range(n_batches):
        n = int(rng.integers(1, per_batch + 1))
        recs = []
        for i in range(n):
            k = int(rng.integers(0, n_keys))
            key = None if rng.random() < keyless else (b"" if k == 0 else
                                                        (b"key-%d" % k) * (9 if long_keys and k % 3 == 0 else 1))
            od = i if not (dup_delta and i % 5 == 4) else i - 1
            hdrs = [C.Header(b"h", b"%d" % i)] if headers and i % 3 == 0 else []
            recs.append(C.Record(0, i & 1, i * 3, od, key, b"v%d-%d" % (b, i), hdrs))
        attributes = 0x20 if control_every and b % control_every == control_every - 1 else 0
        hdr = dict(base_offset=base, last_offset_delta=n - 1, attributes=attributes,
                   base_timestamp=1_700_000_000_000 + b, max_timestamp=1_700_000_000_000 + b + n)
        batch = build_batch(hdr, recs)
        if stored_length_off:
            # the same records with a stored length of 0: Record::decode never uses it
            # (record.rs:259-265), and a rebuilt batch carries the body sizes instead
            full = dict(magic=2, partition_leader_epoch=-1, producer_id=-1)
            full.update(hdr)
            cols, payload, src_off = C.columns_from_records([(full, [C.Record(0, r.attributes, r.timestamp_delta,
                                                                              r.offset_delta, r.key, r.value, r.headers)
                                                                     for r in recs])])
            st, batch, _ = oracle.encode_columns(cols, 1, payload, src_off)
            assert st == C.OK
        out.append(batch)
        base += n
    return out


def model_compact(batches, decode):
    """The reference's compaction in Python; `decode` gives (header dict, records) of a
    deflated batch.  Returns (removed list, action list, rewritten bytes or None list)."""
    n = len(batches)
    removed, action, pieces = [0] * n, [0] * n, [None] * n
    head = set()
    for b in reversed(range(n)):
        hdr, recs = decode(batches[b])
        if hdr["attributes"] & 0x20:
            continue
        last, gone = {}, 0
        for r in recs:
            if r.key is None:
                continue
            if r.key in head:
                gone += 1
                continue
            if r.key in last:
                gone += 1
            last[r.key] = r.offset_delta
        head |= {r.key for r in recs if r.key is not None}
        removed[b] = gone
        if gone:
            retain = set(last.values())
            kept = [r for r in recs if r.key is None or r.offset_delta in retain]
            if not kept:
                action[b] = 2
            else:
                action[b] = 1
                pieces[b] = build_batch(hdr, kept)
    return removed, action, pieces


CASES = {
    "few_keys": dict(n_batches=6, per_batch=40, n_keys=10),
    "many_keys": dict(n_batches=12, per_batch=200, n_keys=5000),
    "all_keyless": dict(n_batches=3, per_batch=20, n_keys=4, keyless=1.0),
    "control_batches": dict(n_batches=9, per_batch=30, n_keys=12, control_every=3),
    "duplicate_offset_deltas": dict(n_batches=5, per_batch=60, n_keys=8, dup_delta=True),
    "headers_and_long_keys": dict(n_batches=7, per_batch=50, n_keys=20, headers=True, long_keys=True),
    "one_key": dict(n_batches=8, per_batch=25, n_keys=1, keyless=0.0),
    "single_batch": dict(n_batches=1, per_batch=300, n_keys=40),
    "stored_lengths_differ": dict(n_batches=4, per_batch=80, n_keys=6, stored_length_off=True),
}
