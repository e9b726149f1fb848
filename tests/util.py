"""Helpers shared by the parity tests."""
import json
import os
import struct

import numpy as np

import oracle
from tansu_amd import columns as C

HERE = os.path.dirname(os.path.abspath(__file__))


def load_vectors():
    with open(os.path.join(HERE, "golden", "vectors.json")) as f:
        return json.load(f)


def find_batches(frame: bytes):
    """Locate RecordBatch v2 batches inside a Kafka frame by their signature:
    magic 2, a batch_length that fits, and a CRC-32C that matches.  Returns a
    list of (start, end) of contiguous batches."""
    out = []
    p = 0
    n = len(frame)
    while p + 61 <= n:
        bl = struct.unpack_from(">i", frame, p + 8)[0]
        if frame[p + 16] == 2 and 49 <= bl <= n - p - 12:
            crc = struct.unpack_from(">I", frame, p + 17)[0]
            if oracle.crc32c(frame[p + 21:p + 12 + bl]) == crc:
                out.append((p, p + 12 + bl))
                p += 12 + bl
                continue
        p += 1
    return out


def record_from_json(r) -> C.Record:
    b = lambda v: None if v is None else bytes(v)
    return C.Record(r["length"], r["attributes"], r["timestamp_delta"], r["offset_delta"],
                    b(r["key"]), b(r["value"]),
                    [C.Header(b(h["key"]), b(h["value"])) for h in r["headers"]])


def offsets_of(batches):
    off = np.zeros(len(batches) + 1, dtype=np.uint64)
    for i, b in enumerate(batches):
        off[i + 1] = off[i] + len(b)
    return off


BATCH_FIELDS = [n for n, _ in C.BATCH_COLUMNS if n not in ("rec_base", "hdr_base", "hdr_count")]


def header_slots(base, count):
    """Indices of every header of every record, in record order."""
    count = np.maximum(count.astype(np.int64), 0)
    total = int(count.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    starts = np.repeat(base.astype(np.int64), count)
    first = np.repeat(np.cumsum(count) - count, count)
    return starts + (np.arange(total, dtype=np.int64) - first)


def assert_same(gpu: dict, cpu: dict, n_batches: int):
    """Compare decoded columns batch by batch.  A failed batch may leave unused
    record slots on the GPU side, and header slots are handed out in runs whose
    order is not defined, so records are located through rec_base and headers
    through each record's rhdr_base."""
    for name in BATCH_FIELDS:
        assert (gpu[name][:n_batches] == cpu[name][:n_batches]).all(), name
    ok = cpu["status"][:n_batches] == C.OK
    assert (gpu["hdr_count"][:n_batches][ok] == cpu["hdr_count"][:n_batches][ok]).all()
    rec_names = [n for n, _ in C.RECORD_COLUMNS if n != "rhdr_base"]
    hdr_names = [n for n, _ in C.HEADER_COLUMNS]
    counts = cpu["record_count"][:n_batches][ok].astype(np.int64)
    gi = header_slots(gpu["rec_base"][:n_batches][ok], counts)   # record indices, GPU side
    ci = header_slots(cpu["rec_base"][:n_batches][ok], counts)
    for name in rec_names:
        assert (gpu[name][gi] == cpu[name][ci]).all(), name
    gh = header_slots(gpu["rhdr_base"][gi], gpu["rhdr_count"][gi])
    ch = header_slots(cpu["rhdr_base"][ci], cpu["rhdr_count"][ci])
    assert len(np.unique(gh)) == len(gh)                      # no slot is shared
    for name in hdr_names:
        assert (gpu[name][gh] == cpu[name][ch]).all(), name


def body_size(rec: C.Record) -> int:
    """record::Builder::size_in_bytes (record.rs:363-372): what Record::try_from(Builder)
    stores in `length`."""
    def octets(b):
        return oracle.varint_size(-1 if b is None else len(b)) + (0 if b is None else len(b))
    n = 1 + oracle.varint_size(rec.timestamp_delta, "i64") + oracle.varint_size(rec.offset_delta)
    n += octets(rec.key) + octets(rec.value) + oracle.varint_size(len(rec.headers))
    for h in rec.headers:
        n += octets(h.key) + octets(h.value)
    return n


def build_batch(header: dict, records) -> bytes:
    """A deflated batch as inflated::Batch::builder()...build() then try_into() gives
    it: every record's length is its body size, record_count, batch_length and crc
    follow from the records (deflated.rs:392-409)."""
    recs = [C.Record(body_size(r), r.attributes, r.timestamp_delta, r.offset_delta, r.key, r.value,
                     r.headers) for r in records]
    hdr = dict(magic=2, partition_leader_epoch=-1, producer_id=-1)
    hdr.update(header)
    cols, payload, src_off = C.columns_from_records([(hdr, recs)])
    # (room by the records' sizes: records of a few bytes outweigh their payload many times)
    st, out, _ = oracle.encode_columns(cols, 1, payload, src_off, out_cap=sum(r.length + 5 for r in recs) + 1024)
    assert st == C.OK
    return out
