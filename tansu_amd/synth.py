"""Deterministic synthetic RecordBatch v2 inputs for tests and bench.py.

Batches have the shape the reference's load generator produces
(nisshi-perf/src/lib.rs:411-431): record i of a batch has offset_delta = i and
timestamp_delta = i, uniformly random payload bytes, magic 2, no compression,
producer fields -1.  Because a record's encoding depends only on its index and
the span sizes, the record_data of a batch with R records is a prefix of the
record_data of a longer batch of the same shape; that is how ragged batch sizes
are built cheaply.

Random bytes come from a counter-based splitmix64 stream (seed + index), so any
slice of any input can be regenerated anywhere, CPU or GPU.

Writing wire bytes here is input generation, not the codec under test: the CRC
field is left zero and filled from the codec's own crc_computed column by the
caller (`patch_crc`), after the CRC kernel has been checked against known
answers.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

SEED_BASE = 0x6E69737368690000  # "nisshi\0\0"
BASE_TIMESTAMP = 1_700_000_000_000


def splitmix64(seed: int, n_words: int, start: int = 0) -> np.ndarray:
    """n_words outputs of splitmix64 seeded with `seed`, starting at output `start`."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + 1 + n_words, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_bytes(seed: int, n: int) -> np.ndarray:
    return splitmix64(seed, (n + 7) // 8).view(np.uint8)[:n]


def _zigzag(v: int) -> int:
    return (v << 1) ^ (v >> 63)


def varint(v: int) -> bytes:
    z = _zigzag(v) & 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while z >= 0x80:
        out.append((z & 0x7F) | 0x80)
        z >>= 7
    out.append(z)
    return bytes(out)


@dataclass(frozen=True)
class Shape:
    """Sizes of the spans of every record; -1 = null."""
    key_len: int = -1
    val_len: int = 100
    n_headers: int = 0
    hkey_len: int = 8
    hval_len: int = 16

    def body(self, i: int) -> int:
        """Bytes of a record whose two deltas are i, without its length varint."""
        n = 1 + len(varint(i)) + len(varint(i))
        for ln in (self.key_len, self.val_len):
            n += len(varint(ln)) + max(ln, 0)
        n += len(varint(self.n_headers))
        n += self.n_headers * (len(varint(self.hkey_len)) + max(self.hkey_len, 0)
                               + len(varint(self.hval_len)) + max(self.hval_len, 0))
        return n


def every_third_missing(i: int) -> int:
    """The deltas of a batch that lost every third record (2, 5, 8, ...) to
    compaction: 0, 1, 3, 4, 6, 7, ..."""
    return i + i // 2


class Template:
    """record_data of a batch of `r_max` records of one shape, payload bytes zero.
    `delta(i)`: the offset delta and timestamp delta of record i (default: i)."""

    def __init__(self, shape: Shape, r_max: int, delta=None):
        self.shape, self.r_max = shape, r_max
        self.delta = delta or (lambda i: i)
        parts = []
        mask_parts = []
        ends = np.zeros(r_max, dtype=np.int64)
        pos = 0
        hdr = (varint(shape.hkey_len), varint(shape.hval_len))
        # records of the same index width share their layout: build per distinct (varint(i) size)
        for i in range(r_max):
            vi = varint(self.delta(i))
            body = shape.body(self.delta(i))
            rec = bytearray()
            m = bytearray()

            def fixed(b):
                rec.extend(b)
                m.extend(b"\0" * len(b))

            def payload(n):
                if n > 0:
                    rec.extend(b"\0" * n)
                    m.extend(b"\1" * n)

            fixed(varint(body))
            fixed(b"\0")
            fixed(vi)
            fixed(vi)
            fixed(varint(shape.key_len))
            payload(shape.key_len)
            fixed(varint(shape.val_len))
            payload(shape.val_len)
            fixed(varint(shape.n_headers))
            for _ in range(shape.n_headers):
                fixed(hdr[0])
                payload(shape.hkey_len)
                fixed(hdr[1])
                payload(shape.hval_len)
            parts.append(bytes(rec))
            mask_parts.append(bytes(m))
            pos += len(rec)
            ends[i] = pos
        self.data = np.frombuffer(b"".join(parts), dtype=np.uint8)
        self.mask = np.frombuffer(b"".join(mask_parts), dtype=np.uint8).astype(bool)
        self.ends = ends

    def data_len(self, r: int) -> int:
        return int(self.ends[r - 1]) if r else 0


def batch_header(base_offset: int, data_len: int, record_count: int, attributes: int = 0,
                 crc: int = 0, last_delta: int = None) -> bytes:
    if last_delta is None:
        last_delta = max(record_count - 1, 0)
    return struct.pack(">qiibIhiqqqhiI", base_offset, 49 + data_len, -1, 2, crc, attributes,
                       last_delta, BASE_TIMESTAMP, BASE_TIMESTAMP + last_delta, -1, -1, -1, record_count)


def make_batches(shape: Shape, counts, seed: int, template: Template = None, delta=None):
    """Build a blob of len(counts) batches; batch b has counts[b] records, record i
    of a batch the deltas delta(i) (default: i).

    Returns (blob np.uint8[], batch_off np.uint64[n + 1], n_records, n_headers).
    CRC fields are zero (see patch_crc).
    """
    counts = np.asarray(counts, dtype=np.int64)
    r_max = int(counts.max()) if len(counts) else 0
    if template is None:
        template = Template(shape, max(r_max, 1), delta)
    delta = template.delta
    lens = np.array([61 + template.data_len(int(r)) for r in counts], dtype=np.int64)
    off = np.zeros(len(counts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    blob = np.zeros(total, dtype=np.uint8)
    base_offset = 0
    uniform = len(counts) > 0 and bool((counts == counts[0]).all())
    if uniform:
        r = int(counts[0])
        dl = template.data_len(r)
        row = np.zeros(61 + dl, dtype=np.uint8)
        row[:61] = np.frombuffer(batch_header(0, dl, r, last_delta=delta(max(r - 1, 0))), dtype=np.uint8)
        row[61:] = template.data[:dl]
        view = blob.reshape(len(counts), 61 + dl)
        view[:] = row
        bo = (np.arange(len(counts), dtype=np.uint64) * np.uint64(delta(max(r - 1, 0)) + 1)).astype(">u8")
        view[:, :8] = bo.view(np.uint8).reshape(len(counts), 8)
        mask = np.zeros(61 + dl, dtype=bool)
        mask[61:] = template.mask[:dl]
        idx = np.flatnonzero(mask)
        rnd = random_bytes(seed, len(counts) * len(idx)).reshape(len(counts), len(idx))
        view[:, idx] = rnd
    else:
        used = 0
        for b, r in enumerate(counts):
            r = int(r)
            dl = template.data_len(r)
            s = int(off[b])
            blob[s:s + 61] = np.frombuffer(batch_header(base_offset, dl, r, last_delta=delta(max(r - 1, 0))), dtype=np.uint8)
            seg = blob[s + 61:s + 61 + dl]
            seg[:] = template.data[:dl]
            idx = np.flatnonzero(template.mask[:dl])
            seg[idx] = splitmix64(seed, (len(idx) + 7) // 8, start=used).view(np.uint8)[:len(idx)]
            used += (len(idx) + 7) // 8
            base_offset += delta(max(r - 1, 0)) + 1
    n_records = int(counts.sum())
    return blob, off, n_records, n_records * shape.n_headers


def _uniform_ints(seed: int, n: int, lo: int, hi: int) -> np.ndarray:
    """n integers uniform in [lo, hi] from the splitmix64 stream of `seed`."""
    return (lo + (splitmix64(seed, n) >> np.uint64(11)) % np.uint64(hi - lo + 1)).astype(np.int64)


def make_var_batches(n_batches: int, records: int, seed: int, key_len: int = 16, val_range=(32, 96),
                     n_headers: int = 2, hkey_len: int = 8, hval_range=(0, 32), delta=None):
    """Batches whose records differ in size from one to the next: every record
    draws its value length from val_range and the value length of each of its
    headers from hval_range (both inclusive), everything else as in make_batches
    (offset_delta = timestamp_delta = index in the batch, or `delta` of it -- a function
    of the array of indices -- for batches that do not number their records; random
    payload bytes).  Returns (blob, batch_off, n_records, n_headers); CRC fields are zero.

    The blob is filled with random bytes first and the structural bytes (lengths,
    counts, deltas, attributes) are written over them, so the whole input is built
    by a handful of array operations."""
    n = n_batches * records
    assert records < (1 << 13) and max(key_len, val_range[1], hkey_len, hval_range[1]) < (1 << 13)

    def zz_size(v):                       # bytes of the zigzag varint of v (v < 2^13, or -1)
        z = np.where(v < 0, 1, 2 * v)
        return np.where(z < 128, 1, 2).astype(np.int64)

    idx = np.tile(np.arange(records, dtype=np.int64), n_batches)
    last_delta = None
    if delta is not None:
        idx = np.asarray(delta(idx), dtype=np.int64)
        last_delta = int(idx[records - 1])
        assert last_delta < (1 << 13)
    vl = _uniform_ints(seed + 1, n, *val_range)
    hvl = _uniform_ints(seed + 2, n * max(n_headers, 1), *hval_range).reshape(n, max(n_headers, 1))[:, :n_headers]
    kl = np.full(n, key_len, dtype=np.int64)
    hkl = np.full(n, hkey_len, dtype=np.int64)
    hc = np.full(n, n_headers, dtype=np.int64)
    # field sizes, in wire order; a negative length is a null span of no bytes
    pieces = [("var", idx), ("var", idx), ("var", kl), ("pay", kl), ("var", vl), ("pay", vl), ("var", hc)]
    for j in range(n_headers):
        pieces += [("var", hkl), ("pay", hkl), ("var", hvl[:, j]), ("pay", hvl[:, j])]
    body = np.ones(n, dtype=np.int64)     # the attributes byte
    for kind, v in pieces:
        body += zz_size(v) if kind == "var" else np.maximum(v, 0)
    rec = zz_size(body) + body
    rec_end = np.cumsum(rec.reshape(n_batches, records), axis=1)
    data_len = rec_end[:, -1]
    off = np.zeros(n_batches + 1, dtype=np.uint64)
    off[1:] = np.cumsum(61 + data_len)
    blob = random_bytes(seed, int(off[-1])).copy()
    starts = (off[:-1].astype(np.int64)[:, None] + 61 + rec_end - rec.reshape(n_batches, records)).reshape(-1)

    def put_var(pos, v):                  # zigzag varints of one or two bytes at pos
        z = np.where(v < 0, 1, 2 * v)
        two = z >= 128
        blob[pos] = np.where(two, (z & 0x7F) | 0x80, z).astype(np.uint8)
        blob[pos[two] + 1] = (z[two] >> 7).astype(np.uint8)
        return pos + np.where(two, 2, 1)

    pos = put_var(starts, body)
    blob[pos] = 0                         # attributes
    pos = pos + 1
    for kind, v in pieces:
        pos = put_var(pos, v) if kind == "var" else pos + np.maximum(v, 0)
    assert (pos == starts + rec).all()
    for b in range(n_batches):
        s = int(off[b])
        blob[s:s + 61] = np.frombuffer(batch_header(b * records, int(data_len[b]), records, last_delta=last_delta),
                                       dtype=np.uint8)
    return blob, off, n, n * n_headers


def patch_crc(blob: np.ndarray, batch_off: np.ndarray, crc: np.ndarray) -> None:
    """Write per-batch CRC values (big-endian) into bytes 17..21 of every batch."""
    be = np.ascontiguousarray(crc, dtype=np.uint32).astype(">u4").view(np.uint8).reshape(-1, 4)
    starts = batch_off[:-1].astype(np.int64) + 17
    for k in range(4):
        blob[starts + k] = be[:, k]


# ---- the named benchmark workloads ---------------------------------------------

C1_SHAPE = Shape(key_len=-1, val_len=100, n_headers=0)
C2_SHAPE = Shape(key_len=16, val_len=64, n_headers=2, hkey_len=8, hval_len=16)
C3_SHAPE = Shape(key_len=-1, val_len=1 << 20, n_headers=0)


def config(name: str, scale: float = 1.0):
    """(shape, counts, seed) of a named workload; `scale` shrinks the batch count."""
    if name == "c1":   # 10k batches x 100 records x 100 B values (the CPU-runnable case)
        return C1_SHAPE, np.full(max(1, int(10_000 * scale)), 100), SEED_BASE + 1
    if name == "c2":   # small records: 16 B keys, 64 B values, 2 headers, 1k records/batch
        return C2_SHAPE, np.full(max(1, int(8_192 * scale)), 1000), SEED_BASE + 2
    if name == "c2v":  # C2 with sizes that change record to record: see make_var_batches
        return None, np.full(max(1, int(8_192 * scale)), 1000), SEED_BASE + 6
    if name == "c3":   # large values: 1 MiB values, 4 records/batch
        return C3_SHAPE, np.full(max(1, int(256 * scale)), 4), SEED_BASE + 3
    if name in ("c5", "c5g"):   # ragged: records per batch log-uniform in [1, 10 000], 100 B values
        # (c5g: the same batches after a compaction that took every third record: see DELTAS)
        n = max(1, int(4_096 * scale))
        u = splitmix64(SEED_BASE + 5, n).astype(np.float64) / 2.0 ** 64
        counts = np.floor(np.exp(u * np.log(10_000.0))).astype(np.int64).clip(1, 10_000)
        return C1_SHAPE, counts, SEED_BASE + 5
    raise ValueError(name)


DELTAS = {"c5g": every_third_missing}   # workloads whose records are not numbered 0, 1, 2 ...: make_batches' `delta`


def mixed_partitions(n_partitions: int, batches_each: int, seed: int = SEED_BASE + 4, only=None):
    """C4: partition p draws its shape from five kinds by p mod 5.  Returns a list
    of (blob, batch_off, n_records, n_headers), one per partition (of the
    partitions in `only`, when given)."""
    kinds = [
        (C1_SHAPE, 100),
        (C2_SHAPE, 1000),
        (Shape(key_len=-1, val_len=1 << 16, n_headers=0), 4),
        (Shape(key_len=-1, val_len=1024, n_headers=0), 1),
        (Shape(key_len=4, val_len=6, n_headers=0), 1),   # control-batch sized record
    ]
    out = []
    for p in (range(n_partitions) if only is None else only):
        shape, r = kinds[p % 5]
        out.append(make_batches(shape, np.full(batches_each, r), seed + 1000 * p))
    return out


def uvarint(v: int) -> bytes:
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def make_produce_requests(blob: np.ndarray, batch_off: np.ndarray, seed: int, max_partitions: int = 8,
                          max_batches: int = 4):
    """The batches of `blob` as they arrive on the wire: ProduceRequest v11 partition
    data, one after the other (requests of 1..max_partitions partitions; a partition's
    data is its index, i32, then the records field, a flexible prefix uvarint(len + 1)
    and 1..max_batches batches, then an empty tagged-field section: message
    ProduceRequest, PartitionProduceData, of the Kafka protocol).  Returns (buf,
    field_pos, batch_off): the buffer, where each records field's prefix starts, and
    the starts of the batches in buf with the end of the last field as last entry."""
    n = len(batch_off) - 1
    draws = splitmix64(seed, 2 * n + 2)
    parts = []
    field_pos, new_off = [], []
    at, b, i = 0, 0, 0
    while b < n:
        in_request = 1 + int(draws[2 * i] % max_partitions)
        for p in range(in_request):
            if b >= n:
                break
            take = min(1 + int(draws[2 * i + 1] % max_batches), n - b)
            i += 1
            lo, hi = int(batch_off[b]), int(batch_off[b + take])
            head = int(p).to_bytes(4, "big")
            prefix = uvarint(hi - lo + 1)
            field_pos.append(at + len(head))
            start = at + len(head) + len(prefix)
            new_off.extend(start + int(batch_off[b + j]) - lo for j in range(take))
            parts.append(head + prefix)
            parts.append(blob[lo:hi].tobytes())
            parts.append(b"\x00")
            at = start + (hi - lo) + 1
            b += take
        i += 1
    new_off.append(at - 1)   # the end of the last field
    buf = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
    return buf, np.asarray(field_pos, dtype=np.uint64), np.asarray(new_off, dtype=np.uint64)
