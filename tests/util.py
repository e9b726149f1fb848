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
