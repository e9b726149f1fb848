"""Column (SoA) layout of decoded record batches, and conversions to and from the
record shape the reference uses.

The layout is the one `include/nisshi_hip.h` declares as `nisshi_columns`: one
array per field, per batch, per record and per header.  Span offsets are
relative to the start of the batch's `record_data`, as `Bytes::split_to` spans
are in the reference (nisshi-sans-io/src/record/codec.rs:65-76); a null span
has length -1 and keeps the offset where its bytes would have started.

Everything here is host-side bookkeeping on numpy arrays: no decoding or
encoding of wire bytes happens in this module.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

# (name, numpy dtype) in the order of the C struct
BATCH_COLUMNS = [
    ("base_offset", np.int64),
    ("batch_length", np.int32),
    ("partition_leader_epoch", np.int32),
    ("magic", np.int8),
    ("crc", np.uint32),
    ("attributes", np.int16),
    ("last_offset_delta", np.int32),
    ("base_timestamp", np.int64),
    ("max_timestamp", np.int64),
    ("producer_id", np.int64),
    ("producer_epoch", np.int16),
    ("base_sequence", np.int32),
    ("record_count", np.uint32),
    ("crc_computed", np.uint32),
    ("data_off", np.uint64),
    ("data_len", np.uint32),
    ("rec_base", np.uint32),
    ("hdr_base", np.uint32),
    ("hdr_count", np.uint32),
    ("status", np.int32),
]
RECORD_COLUMNS = [
    ("length", np.int32),
    ("rattributes", np.uint8),
    ("timestamp_delta", np.int64),
    ("offset_delta", np.int32),
    ("key_off", np.uint32),
    ("key_len", np.int32),
    ("val_off", np.uint32),
    ("val_len", np.int32),
    ("rhdr_base", np.uint32),
    ("rhdr_count", np.int32),
]
HEADER_COLUMNS = [
    ("hkey_off", np.uint32),
    ("hkey_len", np.int32),
    ("hval_off", np.uint32),
    ("hval_len", np.int32),
]
ALL_COLUMNS = BATCH_COLUMNS + RECORD_COLUMNS + HEADER_COLUMNS

# the 13 wire fields of the 61-byte batch header, in wire order
HEADER_FIELDS = [n for n, _ in BATCH_COLUMNS[:13]]

# status codes shared by the C ABI and the CPU restatement
OK, TRY_GET, OVERFLOW, TRY_FROM_INT, UNKNOWN_COMPRESSION, MALFORMED, COMPRESSED, CAPACITY = range(8)
STATUS_NAMES = ["OK", "TRY_GET", "OVERFLOW", "TRY_FROM_INT", "UNKNOWN_COMPRESSION", "MALFORMED",
                "COMPRESSED", "CAPACITY"]

# bytes of decoded output per unit, used for the algorithmic byte count
BATCH_BYTES = 64
RECORD_BYTES = 41
HEADER_BYTES = 16


class CColumns(ctypes.Structure):
    """ctypes view of `nisshi_columns`: one pointer per column."""
    _fields_ = [(name, ctypes.c_void_p) for name, _ in ALL_COLUMNS]


def alloc_host(n_batches: int, n_records: int, n_headers: int) -> dict:
    cols = {}
    for name, dt in BATCH_COLUMNS:
        cols[name] = np.zeros(max(n_batches, 1), dtype=dt)
    for name, dt in RECORD_COLUMNS:
        cols[name] = np.zeros(max(n_records, 1), dtype=dt)
    for name, dt in HEADER_COLUMNS:
        cols[name] = np.zeros(max(n_headers, 1), dtype=dt)
    return cols


def as_c(cols: dict) -> CColumns:
    """Pointers to host numpy arrays (keep `cols` alive while the struct is in use)."""
    c = CColumns()
    for name, _ in ALL_COLUMNS:
        setattr(c, name, cols[name].ctypes.data)
    return c


def trim(cols: dict, n_batches: int, n_records: int, n_headers: int) -> dict:
    out = {}
    for name, _ in BATCH_COLUMNS:
        out[name] = cols[name][:n_batches]
    for name, _ in RECORD_COLUMNS:
        out[name] = cols[name][:n_records]
    for name, _ in HEADER_COLUMNS:
        out[name] = cols[name][:n_headers]
    return out


# ---- the reference's record shape (record.rs:172-198, header.rs:21-29) -------


@dataclass
class Header:
    key: Optional[bytes] = None
    value: Optional[bytes] = None


@dataclass
class Record:
    length: int = 0
    attributes: int = 0
    timestamp_delta: int = 0
    offset_delta: int = 0
    key: Optional[bytes] = None
    value: Optional[bytes] = None
    headers: List[Header] = field(default_factory=list)


def _span(data: memoryview, off: int, length: int) -> Optional[bytes]:
    if length < 0:
        return None
    return bytes(data[off:off + length])


def records_of_batch(cols: dict, b: int, blob) -> List[Record]:
    """Expand batch `b` of decoded columns into the reference's Vec<Record>."""
    view = memoryview(blob)
    base = int(cols["data_off"][b])
    data = view[base:base + int(cols["data_len"][b])]
    out = []
    r0 = int(cols["rec_base"][b])
    for r in range(r0, r0 + int(cols["record_count"][b])):
        h0 = int(cols["rhdr_base"][r])
        headers = [
            Header(_span(data, int(cols["hkey_off"][h]), int(cols["hkey_len"][h])),
                   _span(data, int(cols["hval_off"][h]), int(cols["hval_len"][h])))
            for h in range(h0, h0 + int(cols["rhdr_count"][r]))
        ]
        out.append(Record(
            length=int(cols["length"][r]),
            attributes=int(cols["rattributes"][r]),
            timestamp_delta=int(cols["timestamp_delta"][r]),
            offset_delta=int(cols["offset_delta"][r]),
            key=_span(data, int(cols["key_off"][r]), int(cols["key_len"][r])),
            value=_span(data, int(cols["val_off"][r]), int(cols["val_len"][r])),
            headers=headers,
        ))
    return out


def columns_from_records(batches: list) -> tuple:
    """Build encode input from the reference's shape.

    `batches` is a list of (header_fields: dict, records: List[Record]).  Returns
    (cols, payload: bytes, src_data_off: np.uint64[n]): the spans of batch b live
    in `payload` at src_data_off[b] + offset.
    """
    n_b = len(batches)
    n_r = sum(len(recs) for _, recs in batches)
    n_h = sum(len(r.headers) for _, recs in batches for r in recs)
    cols = alloc_host(n_b, n_r, n_h)
    payload = bytearray()
    src_off = np.zeros(max(n_b, 1), dtype=np.uint64)
    r = h = 0
    for b, (hdr, recs) in enumerate(batches):
        for name in HEADER_FIELDS:
            if name in hdr:
                cols[name][b] = hdr[name]
        cols["record_count"][b] = len(recs)
        cols["rec_base"][b] = r
        cols["hdr_base"][b] = h
        cols["status"][b] = OK
        base = len(payload)
        src_off[b] = base
        h_start = h

        def put(data):
            off = len(payload) - base
            if data is None:
                return off, -1
            payload.extend(data)
            return off, len(data)

        for rec in recs:
            cols["length"][r] = rec.length
            cols["rattributes"][r] = rec.attributes
            cols["timestamp_delta"][r] = rec.timestamp_delta
            cols["offset_delta"][r] = rec.offset_delta
            cols["key_off"][r], cols["key_len"][r] = put(rec.key)
            cols["val_off"][r], cols["val_len"][r] = put(rec.value)
            cols["rhdr_base"][r] = h
            cols["rhdr_count"][r] = len(rec.headers)
            for hd in rec.headers:
                cols["hkey_off"][h], cols["hkey_len"][h] = put(hd.key)
                cols["hval_off"][h], cols["hval_len"][h] = put(hd.value)
                h += 1
            r += 1
        cols["hdr_count"][b] = h - h_start
        cols["data_off"][b] = base
        cols["data_len"][b] = len(payload) - base
    return cols, bytes(payload), src_off[:n_b] if n_b else src_off[:0]


def algorithmic_decode_bytes(n_batches: int, n_records: int, n_headers: int, in_bytes: int) -> int:
    """Bytes a decode must move at minimum: every input byte read once (the CRC
    needs them all) plus the decoded columns written once."""
    return in_bytes + BATCH_BYTES * n_batches + RECORD_BYTES * n_records + HEADER_BYTES * n_headers
