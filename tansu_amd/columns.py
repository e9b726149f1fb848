"""Columns of decoded record batches, and conversions to and from the record
shape the reference uses.

Host code here works on one numpy array per field ("fields": BATCH_COLUMNS,
RECORD_COLUMNS, HEADER_COLUMNS).  The device library stores the per-record and
per-header fields in groups of 16 bytes, as `include/nisshi_hip.h` declares in
`nisshi_columns` (GROUPS below); `pack` and `unpack` convert between the two.
Span offsets are relative to the start of the batch's `record_data`, as
`Bytes::split_to` spans are in the reference
(nisshi-sans-io/src/record/codec.rs:65-76); a null span has length -1 and keeps
the offset where its bytes would have started.

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


# The groups of `nisshi_columns`: (array name, fields stored together as four
# 32-bit words).  timestamp_delta and rattributes are arrays of their own.
RECORD_GROUPS = [
    ("rkey", ["length", "offset_delta", "key_off", "key_len"]),
    ("rval", ["val_off", "val_len", "rhdr_base", "rhdr_count"]),
]
HEADER_GROUPS = [
    ("hdr", ["hkey_off", "hkey_len", "hval_off", "hval_len"]),
]
# arrays of the device layout, in the order of the C struct: (name, dtype, words per row, level)
DEVICE_ARRAYS = ([(n, dt, 1, 0) for n, dt in BATCH_COLUMNS] +
                 [("rkey", np.int32, 4, 1), ("rval", np.int32, 4, 1),
                  ("timestamp_delta", np.int64, 1, 1), ("rattributes", np.uint8, 1, 1),
                  ("hdr", np.int32, 4, 2),
                  # a batch's record data in bytes plus one where the decode found every varint
                  # written the shortest way, else zero: what the encoder need not size again
                  ("enc_len", np.uint32, 1, 0)])
_FIELD_DTYPE = dict(ALL_COLUMNS)


class CColumns(ctypes.Structure):
    """ctypes view of `nisshi_columns`: one pointer per array of the device layout."""
    _fields_ = [(name, ctypes.c_void_p) for name, _, _, _ in DEVICE_ARRAYS]


def alloc_device_layout(n_batches: int, n_records: int, n_headers: int) -> dict:
    """Host arrays in the device layout (for the host-buffer entry points)."""
    counts = (n_batches, n_records, n_headers)
    out = {}
    for name, dt, words, level in DEVICE_ARRAYS:
        n = max(counts[level], 1)
        out[name] = np.zeros((n, words) if words > 1 else n, dtype=dt)
    return out


def as_c(arrays: dict) -> CColumns:
    """Pointers to host arrays in the device layout (keep `arrays` alive while in use)."""
    c = CColumns()
    for name, _, _, _ in DEVICE_ARRAYS:
        setattr(c, name, arrays[name].ctypes.data)
    return c


def pack(cols: dict) -> dict:
    """Fields -> device layout."""
    out = {n: np.ascontiguousarray(cols[n]) for n, _ in BATCH_COLUMNS}
    for name, fields in RECORD_GROUPS + HEADER_GROUPS:
        n = len(cols[fields[0]])
        g = np.zeros((max(n, 1), 4), dtype=np.int32)
        for i, f in enumerate(fields):
            g[:n, i] = np.asarray(cols[f]).view(np.int32) if np.asarray(cols[f]).dtype.itemsize == 4 \
                else np.asarray(cols[f]).astype(np.int32)
        out[name] = g
    out["timestamp_delta"] = np.ascontiguousarray(cols["timestamp_delta"])
    out["rattributes"] = np.ascontiguousarray(cols["rattributes"])
    n = len(cols["status"])
    out["enc_len"] = np.ascontiguousarray(cols["enc_len"]) if "enc_len" in cols else np.zeros(max(n, 1), np.uint32)
    return out


def unpack(arrays: dict) -> dict:
    """Device layout -> fields."""
    out = {n: arrays[n] for n, _ in BATCH_COLUMNS}
    for name, fields in RECORD_GROUPS + HEADER_GROUPS:
        g = np.asarray(arrays[name]).reshape(-1, 4)
        for i, f in enumerate(fields):
            out[f] = np.ascontiguousarray(g[:, i]).view(_FIELD_DTYPE[f])
    out["timestamp_delta"] = arrays["timestamp_delta"]
    out["rattributes"] = arrays["rattributes"]
    if "enc_len" in arrays:
        out["enc_len"] = arrays["enc_len"]
    return out


def alloc_host(n_batches: int, n_records: int, n_headers: int) -> dict:
    cols = {}
    for name, dt in BATCH_COLUMNS:
        cols[name] = np.zeros(max(n_batches, 1), dtype=dt)
    for name, dt in RECORD_COLUMNS:
        cols[name] = np.zeros(max(n_records, 1), dtype=dt)
    for name, dt in HEADER_COLUMNS:
        cols[name] = np.zeros(max(n_headers, 1), dtype=dt)
    return cols


def trim(cols: dict, n_batches: int, n_records: int, n_headers: int) -> dict:
    out = {}
    for name, _ in BATCH_COLUMNS:
        out[name] = cols[name][:n_batches]
    for name, _ in RECORD_COLUMNS:
        out[name] = cols[name][:n_records]
    for name, _ in HEADER_COLUMNS:
        out[name] = cols[name][:n_headers]
    if "enc_len" in cols:
        out["enc_len"] = cols["enc_len"][:n_batches]
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
