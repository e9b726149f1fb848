"""ctypes binding of the CPU restatement in oracle/ (test infrastructure only)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from tansu_amd import columns as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libnisshi_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libnisshi_oracle.so"])


class FieldColumns(ctypes.Structure):
    """ctypes view of the oracle's `no_columns`: one pointer per field."""
    _fields_ = [(name, ctypes.c_void_p) for name, _ in C.ALL_COLUMNS]


def as_c(cols: dict) -> FieldColumns:
    """Pointers to one numpy array per field (keep `cols` alive while in use)."""
    c = FieldColumns()
    for name, _ in C.ALL_COLUMNS:
        setattr(c, name, cols[name].ctypes.data)
    return c


def _load():
    src = os.path.join(ORACLE_DIR, "nisshi_oracle.c")
    if not os.path.exists(LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB_PATH)):
        build()
    lib = ctypes.CDLL(LIB_PATH)
    u8p = ctypes.c_char_p
    sz = ctypes.c_size_t
    lib.no_crc32c.restype = ctypes.c_uint32
    lib.no_crc32c.argtypes = [u8p, sz]
    for name in ("no_varint_size",):
        getattr(lib, name).restype = sz
        getattr(lib, name).argtypes = [ctypes.c_int32]
    lib.no_varlong_size.restype = sz
    lib.no_varlong_size.argtypes = [ctypes.c_int64]
    lib.no_uvarint_size.restype = sz
    lib.no_uvarint_size.argtypes = [ctypes.c_uint32]
    lib.no_octets_size.restype = sz
    lib.no_octets_size.argtypes = [ctypes.c_int64]
    lib.no_varint_encode.restype = sz
    lib.no_varint_encode.argtypes = [ctypes.c_int32, ctypes.c_void_p]
    lib.no_varlong_encode.restype = sz
    lib.no_varlong_encode.argtypes = [ctypes.c_int64, ctypes.c_void_p]
    lib.no_uvarint_encode.restype = sz
    lib.no_uvarint_encode.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
    lib.no_varint_decode.argtypes = [u8p, sz, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_int32)]
    lib.no_varlong_decode.argtypes = [u8p, sz, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_int64)]
    lib.no_uvarint_decode.argtypes = [u8p, sz, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint32)]
    lib.no_compression.argtypes = [ctypes.c_int16, ctypes.POINTER(ctypes.c_int)]
    lib.no_attributes_clear_compression.argtypes = [ctypes.c_int16, ctypes.POINTER(ctypes.c_int16)]
    lib.no_frame_split.argtypes = [u8p, sz, ctypes.c_void_p, sz, ctypes.POINTER(sz)]
    lib.no_records_prefix_decode.argtypes = [u8p, sz, ctypes.c_int, ctypes.POINTER(sz),
                                             ctypes.POINTER(ctypes.c_uint32),
                                             ctypes.POINTER(ctypes.c_int)]
    lib.no_records_prefix_encode.restype = sz
    lib.no_records_prefix_encode.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    lib.no_decode_columns.argtypes = [ctypes.c_void_p, ctypes.c_void_p, sz, ctypes.c_int,
                                      ctypes.POINTER(FieldColumns), sz, sz,
                                      ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.no_decode_columns_lean.argtypes = lib.no_decode_columns.argtypes
    lib.no_encode_columns.argtypes = [ctypes.POINTER(FieldColumns), sz, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, sz, ctypes.c_void_p]
    lib.no_builder_columns.argtypes = [ctypes.POINTER(FieldColumns), sz, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.no_produce_rows.argtypes = [vp, sz, i64, ctypes.c_int, sz, sz, vp] + [vp] * 11 + \
        [ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.no_combine.argtypes = [vp, vp, sz, vp, sz, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_int)]
    lib.no_compact_partition.argtypes = [vp, vp, sz, vp, sz, vp, vp, vp]
    lib.no_fetch.argtypes = [sz] + [vp] * 13 + [i64, vp, sz, vp, sz, ctypes.POINTER(sz)]
    lib.no_snappy_inflate.argtypes = [u8p, sz, vp, sz, ctypes.POINTER(sz)]
    lib.no_inflate_frame.argtypes = [vp, vp, sz, vp, sz, vp, vp]
    lib.no_lz4_frame.argtypes = [u8p, sz, vp, sz, ctypes.POINTER(sz)]
    lib.no_zstd_frames.argtypes = [u8p, sz, vp, sz, ctypes.POINTER(sz)]
    lib.no_zstd_room.argtypes = [u8p, sz]
    lib.no_zstd_room.restype = sz
    lib.no_xxh64.argtypes = [u8p, sz, ctypes.c_uint64]
    lib.no_xxh64.restype = ctypes.c_uint64
    lib.no_gzip_member.argtypes = [u8p, sz, vp, sz, ctypes.POINTER(sz)]
    lib.no_xxh32.argtypes = [u8p, sz, ctypes.c_uint32]
    lib.no_xxh32.restype = ctypes.c_uint32
    return lib


LIB = _load()


def snappy_inflate(data: bytes):
    """(status, inflated bytes) of the record data of a Snappy batch."""
    data = bytes(data)
    out = ctypes.create_string_buffer(max(1, 64 * len(data) + 64))
    n = ctypes.c_size_t()
    st = LIB.no_snappy_inflate(data, len(data), out, len(out), ctypes.byref(n))
    return st, out.raw[:n.value]


def gzip_inflate(data: bytes):
    """(status, inflated bytes) of the record data of a gzip batch (one member)."""
    data = bytes(data)
    out = ctypes.create_string_buffer(max(1, 1040 * len(data) + 64))
    n = ctypes.c_size_t()
    st = LIB.no_gzip_member(data, len(data), out, len(out), ctypes.byref(n))
    return st, out.raw[:n.value]


def lz4_inflate(data: bytes):
    """(status, inflated bytes) of the record data of an LZ4 batch (one frame)."""
    data = bytes(data)
    out = ctypes.create_string_buffer(max(1, 300 * len(data) + (8 << 20)))
    n = ctypes.c_size_t()
    st = LIB.no_lz4_frame(data, len(data), out, len(out), ctypes.byref(n))
    return st, out.raw[:n.value]


def zstd_inflate(data: bytes):
    """(status, inflated bytes) of the record data of a zstd batch (its frames, one after the other)."""
    data = bytes(data)
    room = LIB.no_zstd_room(data, len(data))
    if room == 0:
        return 11, b""
    out = ctypes.create_string_buffer(room)
    n = ctypes.c_size_t()
    st = LIB.no_zstd_frames(data, len(data), out, room, ctypes.byref(n))
    return st, out.raw[:n.value]


def xxh64(data: bytes, seed: int = 0) -> int:
    return LIB.no_xxh64(bytes(data), len(data), seed)


def xxh32(data: bytes, seed: int = 0) -> int:
    return LIB.no_xxh32(bytes(data), len(data), seed)


def inflate_frame(blob, batch_off, out_cap=None):
    """(status, out bytes, out_off, per-batch status) of a frame with its Snappy and LZ4 batches inflated."""
    blob = np.ascontiguousarray(np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob)
    off = np.ascontiguousarray(batch_off, dtype=np.uint64)
    n_b = len(off) - 1
    grow = out_cap is None
    out_cap = 1040 * len(blob) + 64 if out_cap is None else out_cap
    while True:
        out = np.zeros(max(out_cap, 1), dtype=np.uint8)
        out_off = np.zeros(n_b + 1, dtype=np.uint64)
        status = np.zeros(max(n_b, 1), dtype=np.int32)
        st = LIB.no_inflate_frame(blob.ctypes.data, off.ctypes.data, n_b, out.ctypes.data, out_cap,
                                  out_off.ctypes.data, status.ctypes.data)
        # (a zstd frame of repeated bytes can hold tens of thousands of bytes a byte)
        if st != 7 or not grow or out_cap >= (1 << 31):
            break
        out_cap *= 8
    return st, out[:int(out_off[n_b])].tobytes() if st == 0 else b"", out_off, status[:n_b]


def crc32c(data: bytes) -> int:
    return LIB.no_crc32c(bytes(data), len(data))


def varint_decode(data: bytes, kind="i32"):
    """Returns (status, value, consumed)."""
    pos = ctypes.c_size_t(0)
    data = bytes(data)
    if kind == "i32":
        out = ctypes.c_int32()
        st = LIB.no_varint_decode(data, len(data), ctypes.byref(pos), ctypes.byref(out))
    elif kind == "i64":
        out = ctypes.c_int64()
        st = LIB.no_varlong_decode(data, len(data), ctypes.byref(pos), ctypes.byref(out))
    else:
        out = ctypes.c_uint32()
        st = LIB.no_uvarint_decode(data, len(data), ctypes.byref(pos), ctypes.byref(out))
    return st, out.value, pos.value


def varint_encode(v: int, kind="i32") -> bytes:
    buf = ctypes.create_string_buffer(16)
    fn = {"i32": LIB.no_varint_encode, "i64": LIB.no_varlong_encode, "u32": LIB.no_uvarint_encode}[kind]
    n = fn(v, buf)
    return buf.raw[:n]


def varint_size(v: int, kind="i32") -> int:
    fn = {"i32": LIB.no_varint_size, "i64": LIB.no_varlong_size, "u32": LIB.no_uvarint_size}[kind]
    return fn(v)


def frame_split(blob: bytes):
    """Returns (status, offsets np.uint64[n+1])."""
    cap = len(blob) // 12 + 2
    off = np.zeros(cap, dtype=np.uint64)
    n = ctypes.c_size_t(0)
    st = LIB.no_frame_split(bytes(blob), len(blob), off.ctypes.data, cap, ctypes.byref(n))
    return st, off[:n.value + 1].copy()


def decode_columns(blob, batch_off, by_value=True, rec_cap=None, hdr_cap=None, lean=False):
    """Decode a blob of batches.  Returns (status, cols, n_records, n_headers).
    `lean`: by the variant that fills the columns without the intermediate Vec<Record>."""
    blob = np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
    batch_off = np.ascontiguousarray(batch_off, dtype=np.uint64)
    n_b = len(batch_off) - 1
    if rec_cap is None:
        rec_cap = len(blob) // 7 + 1
    if hdr_cap is None:
        hdr_cap = len(blob) // 2 + 1
    cols = C.alloc_host(n_b, rec_cap, hdr_cap)
    cc = as_c(cols)
    nr, nh = ctypes.c_size_t(0), ctypes.c_size_t(0)
    fn = LIB.no_decode_columns_lean if lean else LIB.no_decode_columns
    st = fn(blob.ctypes.data, batch_off.ctypes.data, n_b, int(by_value), ctypes.byref(cc), rec_cap, hdr_cap,
            ctypes.byref(nr), ctypes.byref(nh))
    return st, C.trim(cols, n_b, nr.value, nh.value), nr.value, nh.value


def encode_columns(cols, n_batches, src, src_data_off, out_cap=None):
    """Re-encode batches.  Returns (status, out bytes, out_off np.uint64[n+1])."""
    src = np.frombuffer(bytes(src), dtype=np.uint8) if not isinstance(src, np.ndarray) else src
    src_data_off = np.ascontiguousarray(src_data_off, dtype=np.uint64)
    if out_cap is None:
        out_cap = len(src) * 2 + 128 * n_batches + 1024
    out = np.zeros(out_cap, dtype=np.uint8)
    out_off = np.zeros(n_batches + 1, dtype=np.uint64)
    cc = as_c(cols)
    st = LIB.no_encode_columns(ctypes.byref(cc), n_batches, src.ctypes.data,
                               src_data_off.ctypes.data, out.ctypes.data, out_cap,
                               out_off.ctypes.data)
    return st, out[:int(out_off[n_batches])].tobytes(), out_off


def builder_columns(cols, n_batches, src, src_data_off):
    """inflated::Builder::build batch_length and crc per batch (records' length
    is recomputed from the body, as Record::try_from(Builder) does)."""
    src = np.frombuffer(bytes(src), dtype=np.uint8) if not isinstance(src, np.ndarray) else src
    src_data_off = np.ascontiguousarray(src_data_off, dtype=np.uint64)
    bl = np.zeros(max(n_batches, 1), dtype=np.int32)
    crc = np.zeros(max(n_batches, 1), dtype=np.uint32)
    cc = as_c(cols)
    st = LIB.no_builder_columns(ctypes.byref(cc), n_batches, src.ctypes.data,
                                src_data_off.ctypes.data, bl.ctypes.data, crc.ctypes.data)
    return st, bl[:n_batches], crc[:n_batches]


ROW_FIELDS = [("offset", np.int64), ("timestamp", np.int64), ("key_pos", np.uint64),
              ("key_len", np.int64), ("val_pos", np.uint64), ("val_len", np.int64)]
HDR_ROW_FIELDS = [("offset", np.int64), ("key_pos", np.uint64), ("key_len", np.int64),
                  ("val_pos", np.uint64), ("val_len", np.int64)]


def produce_rows(batch: bytes, high: int, check_time: bool = True):
    """The rows pg.rs:797-943 writes for one deflated batch.  Returns (status,
    attributes, rows: dict, header rows: dict); spans are positions in `batch`."""
    buf = np.frombuffer(bytes(batch), dtype=np.uint8)
    cap_r, cap_h = len(buf) // 7 + 1, len(buf) // 2 + 1
    rows = {n: np.zeros(cap_r, dtype=dt) for n, dt in ROW_FIELDS}
    hrows = {n: np.zeros(cap_h, dtype=dt) for n, dt in HDR_ROW_FIELDS}
    attributes = np.zeros(1, dtype=np.int16)
    nr, nh = ctypes.c_size_t(0), ctypes.c_size_t(0)
    st = LIB.no_produce_rows(buf.ctypes.data if len(buf) else None, len(buf), high, int(check_time),
                             cap_r, cap_h, attributes.ctypes.data,
                             *[rows[n].ctypes.data for n, _ in ROW_FIELDS],
                             *[hrows[n].ctypes.data for n, _ in HDR_ROW_FIELDS],
                             ctypes.byref(nr), ctypes.byref(nh))
    return (st, int(attributes[0]), {n: v[:nr.value] for n, v in rows.items()},
            {n: v[:nh.value] for n, v in hrows.items()})


def combine(batches):
    """combine (batch.rs:661-709) of a list of deflated batches.  Returns (status,
    bytes or None)."""
    bufs = [np.frombuffer(bytes(b), dtype=np.uint8) if len(b) else np.zeros(0, dtype=np.uint8)
            for b in batches]
    n = len(bufs)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    lens = (ctypes.c_size_t * max(n, 1))(*[len(b) for b in bufs])
    cap = 3 * sum(len(b) for b in bufs) + 64     # rebased deltas may need longer varints
    out = np.zeros(cap, dtype=np.uint8)
    out_len, none = ctypes.c_size_t(0), ctypes.c_int(0)
    st = LIB.no_combine(ptrs, lens, n, out.ctypes.data, cap, ctypes.byref(out_len), ctypes.byref(none))
    if st or none.value:
        return st, None
    return st, out[:out_len.value].tobytes()


FETCH_FIELDS = [("offset", np.int64), ("attributes", np.int16), ("timestamp", np.int64),
                ("producer_id", np.int64), ("producer_epoch", np.int16), ("completed", np.uint8),
                ("key_off", np.uint32), ("key_len", np.int32), ("val_off", np.uint32),
                ("val_len", np.int32), ("hdr_base", np.uint32), ("hdr", np.uint32)]


def fetch(rows: dict, payload: bytes, now: int, batch_cap: int = 64):
    """The fetch seam (pg.rs:1875-2052; `completed` None: lite.rs:2765-2980) over
    rows given as arrays.  Returns (status, list of deflated batches)."""
    n = len(rows["offset"])
    keep = []
    ptrs = []
    for name, dt in FETCH_FIELDS:
        v = rows.get(name)
        if v is None:
            ptrs.append(None)
            continue
        a = np.ascontiguousarray(v, dtype=dt)
        if a.size == 0:
            a = np.zeros(4, dtype=dt)
        keep.append(a)
        ptrs.append(a.ctypes.data)
    pl = np.frombuffer(bytes(payload), dtype=np.uint8) if len(payload) else np.zeros(1, dtype=np.uint8)
    n_hdr = int(rows["hdr_base"][n]) if n else 0
    cap = len(pl) + 61 * batch_cap + 64 * n + 32 * n_hdr + 1024
    out = np.zeros(cap, dtype=np.uint8)
    out_off = np.zeros(batch_cap + 1, dtype=np.uint64)
    nb = ctypes.c_size_t(0)
    st = LIB.no_fetch(n, *ptrs, pl.ctypes.data, now, out.ctypes.data, cap, out_off.ctypes.data,
                      batch_cap, ctypes.byref(nb))
    return st, [out[int(out_off[b]):int(out_off[b + 1])].tobytes() for b in range(nb.value)]


UNCHANGED, REWRITTEN, DELETED = 0, 1, 2


def compact_partition(batches):
    """Log compaction of one partition's deflated batches, oldest first (slate/
    storage.rs:283-312).  Returns (status, removed np.int64[n], action np.int32[n],
    list of rewritten batches' bytes or None)."""
    bufs = [np.frombuffer(bytes(b), dtype=np.uint8) if len(b) else np.zeros(0, dtype=np.uint8)
            for b in batches]
    n = len(bufs)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    lens = (ctypes.c_size_t * max(n, 1))(*[len(b) for b in bufs])
    total = sum(len(b) for b in bufs)
    cap = total + total // 7 * 4 + 64            # a rebuilt length may be up to four bytes longer
    out = np.zeros(cap, dtype=np.uint8)
    out_off = np.zeros(n + 1, dtype=np.uint64)
    removed = np.zeros(max(n, 1), dtype=np.int64)
    action = np.zeros(max(n, 1), dtype=np.int32)
    st = LIB.no_compact_partition(ptrs, lens, n, out.ctypes.data, cap, out_off.ctypes.data,
                                  removed.ctypes.data, action.ctypes.data)
    pieces = [out[int(out_off[b]):int(out_off[b + 1])].tobytes() if action[b] == REWRITTEN else None
              for b in range(n)]
    return st, removed[:n], action[:n], pieces
