"""Rows for the fetch seam tests and a Python model of the reference's loop
(nisshi-storage/src/pg.rs:1875-2052), independent of the C restatement."""
import numpy as np

from tansu_amd import columns as C
from util import build_batch

NOW = 1_700_000_123_456


def make_rows(n, seed=0, change_every=0, open_rate=0.0, gap_rate=0.1, headers=2, null_rate=0.1,
              with_completed=True, big_gap_at=None):
    """n rows in offset order.  Returns (rows dict of arrays, payload bytes)."""
    rng = np.random.default_rng(seed)
    offset = 1000 + np.cumsum(1 + (rng.random(n) < gap_rate) * rng.integers(1, 5, size=n))
    if big_gap_at is not None and n:
        offset[big_gap_at:] += 1 << 33
    group = np.zeros(n, dtype=np.int64) if not change_every else np.arange(n) // change_every
    attributes = np.where(group % 3 == 1, 0x10, 0).astype(np.int16)
    producer_id = np.where(group % 2 == 0, -1, 7000 + group).astype(np.int64)
    producer_epoch = np.where(group % 2 == 0, -1, group % 5).astype(np.int16)
    timestamp = 1_700_000_000_000 + np.cumsum(rng.integers(-2, 50, size=n))
    completed = (rng.random(n) >= open_rate).astype(np.uint8)
    payload = bytearray()

    def span(lo, hi):
        r = rng.random()
        if r < null_rate:
            return len(payload), -1
        ln = 0 if r < 2 * null_rate else int(rng.integers(lo, hi))
        at = len(payload)
        payload.extend(rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes())
        return at, ln

    key_off, key_len, val_off, val_len = (np.zeros(n, dtype=np.int64) for _ in range(4))
    hdr_base = np.zeros(n + 1, dtype=np.uint32)
    hdr = []
    for i in range(n):
        key_off[i], key_len[i] = span(1, 24)
        val_off[i], val_len[i] = span(1, 700 if i % 17 == 0 else 120)
        hc = int(rng.integers(0, headers + 1)) if headers else 0
        for _ in range(hc):
            ko, kl = span(1, 12)
            vo, vl = span(1, 40)
            hdr.append((ko, kl & 0xffffffff, vo, vl & 0xffffffff))
        hdr_base[i + 1] = hdr_base[i] + hc
    rows = dict(offset=offset.astype(np.int64), attributes=attributes, timestamp=timestamp.astype(np.int64),
                producer_id=producer_id, producer_epoch=producer_epoch,
                completed=completed if with_completed else None,
                key_off=key_off.astype(np.uint32), key_len=key_len.astype(np.int32),
                val_off=val_off.astype(np.uint32), val_len=val_len.astype(np.int32),
                hdr_base=hdr_base, hdr=np.array(hdr, dtype=np.uint32).reshape(-1, 4))
    return rows, bytes(payload)


def model_fetch(rows, payload, now=NOW):
    """The loop of pg.rs:1875-2052 in Python.  Returns a list of deflated batches, or
    None where the reference returns Err (an offset delta that does not fit i32)."""
    n = len(rows["offset"])

    def span(off, ln):
        return None if ln < 0 else payload[int(off):int(off) + int(ln)]

    def builder(i):
        return dict(base_offset=int(rows["offset"][i]), attributes=int(rows["attributes"][i]),
                    base_timestamp=int(rows["timestamp"][i]), producer_id=int(rows["producer_id"][i]),
                    producer_epoch=int(rows["producer_epoch"][i]), max_timestamp=now,
                    base_sequence=0, last_offset_delta=0)

    if n == 0:
        return [build_batch(dict(base_timestamp=now, max_timestamp=now, producer_epoch=0, base_sequence=0), [])]
    out, b, recs, previous = [], builder(0), [], None
    for i in range(n):
        key3 = (int(rows["attributes"][i]), int(rows["producer_id"][i]), int(rows["producer_epoch"][i]))
        if key3 != (b["attributes"], b["producer_id"], b["producer_epoch"]):
            out.append(build_batch(b, recs))
            b, recs = builder(i), []
        offset = int(rows["offset"][i])
        if rows.get("completed") is not None and not rows["completed"][i] and \
                (previous is None or previous + 1 != offset):
            break
        delta = offset - b["base_offset"]
        if not -(1 << 31) <= delta < (1 << 31):
            return None
        h0, h1 = int(rows["hdr_base"][i]), int(rows["hdr_base"][i + 1])
        headers = [C.Header(span(h[0], np.int32(h[1])), span(h[2], np.int32(h[3])))
                   for h in rows["hdr"][h0:h1].astype(np.int64).astype(np.uint32).view(np.int32).reshape(-1, 4)]
        recs.append(C.Record(0, 0, int(rows["timestamp"][i]) - b["base_timestamp"], delta,
                             span(rows["key_off"][i], rows["key_len"][i]),
                             span(rows["val_off"][i], rows["val_len"][i]), headers))
        previous = offset
        b["last_offset_delta"] = delta
    out.append(build_batch(b, recs))
    return out


CASES = {
    "one_batch": dict(n=50),
    "producers_change": dict(n=200, change_every=17),
    "open_transaction": dict(n=120, change_every=40, open_rate=0.05, gap_rate=0.3),
    "open_first_row": dict(n=5, open_rate=1.0),
    "no_headers": dict(n=64, headers=0, null_rate=0.0),
    "lite_engine": dict(n=90, change_every=30, open_rate=0.5, with_completed=False),
    "single_row": dict(n=1),
    "change_each_row": dict(n=33, change_every=1),
    "delta_overflow": dict(n=40, big_gap_at=25),
    "delta_overflow_after_stop": dict(n=40, big_gap_at=25, open_rate=0.2, gap_rate=0.9),
}
