"""Parity of the gzip inflate (nisshi-sans-io/src/lib.rs:1716 in Vec<Record>::try_from,
nisshi-sans-io/src/record/deflated.rs:444-457) with its CPU restatement, through the C ABI: byte-exact output
frames, statuses, and the records a decode of the output gives.  The members come from Python's zlib."""
import struct

import numpy as np
import pytest

import gzip_cases
import oracle
from tansu_amd import columns as C, hip
from test_gpu_inflate import check, snappy_batch
from test_gpu_inflate_lz4 import lz4_batch
from test_gpu_parity import gpu_decode
from test_oracle_gzip import OPTIONS, data_of
from util import build_batch, load_vectors, offsets_of, record_from_json

pytestmark = pytest.mark.gpu
V = load_vectors()


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def framed_batch(batch: bytes, data: bytes) -> bytes:
    head = bytearray(batch[:61])
    struct.pack_into(">i", head, 8, 49 + len(data))
    struct.pack_into(">h", head, 21, (struct.unpack_from(">h", head, 21)[0] & ~7) | 1)
    struct.pack_into(">I", head, 17, oracle.crc32c(bytes(head[21:]) + data))
    return bytes(head) + data


def gzip_batch(batch: bytes, **opts) -> bytes:
    """An uncompressed batch as the gzip batch of the same records (the CRC recomputed)."""
    return framed_batch(batch, gzip_cases.member(batch[61:], **opts))


def test_reference_vector(codec):
    case = [c for c in V["compressed_batches"] if c["name"] == "decode_gzip"][0]
    batch = bytes(case["batch"])
    batch = batch[:12 + struct.unpack_from(">i", batch, 8)[0]]
    got, got_off, status = check(codec, [batch])
    assert int(status[0]) == C.OK
    cols, totals, _, _ = gpu_decode(codec, got, got_off)
    assert C.records_of_batch(cols.to_host(), 0, got) == [record_from_json(r) for r in case["records"]]


@pytest.mark.parametrize("kind", ["text", "random", "zeros", "words"])
def test_members_from_zlib(codec, kind):
    rng = np.random.default_rng(len(kind) + 60)
    batches, plains = [], []
    for n in (1, 5, 300, 5000, 70000, 200000):
        value = data_of(kind, n, rng)
        plain = build_batch(dict(base_offset=n), [C.Record(0, 0, 0, 0, b"k", value), C.Record(0, 0, 1, 1, None, value[:9])])
        for opts in OPTIONS:
            batches.append(gzip_batch(plain, **opts))
            plains.append(plain)
    got, got_off, status = check(codec, batches)
    assert all(int(s) == C.OK for s in status)
    for i, plain in enumerate(plains):
        a = got[int(got_off[i]):int(got_off[i + 1])]
        assert a[:17] == plain[:17] and a[21:] == plain[21:]


def test_members_that_do_not_inflate_and_a_mixed_frame(codec):
    plain = build_batch(dict(base_offset=3), [C.Record(0, 0, i, i, b"key%d" % i, b"value" * i) for i in range(40)])
    zstd = bytes([c for c in V["compressed_batches"] if c["name"] == "decode_zstd"][0]["batch"])
    batches = [plain, gzip_batch(plain), zstd, snappy_batch(plain), lz4_batch(plain), plain[:30], gzip_batch(plain, level=0)]
    names = []
    for name, m in gzip_cases.corrupt_cases().items():
        batches.append(framed_batch(plain, m))
        names.append(name)
    batches.append(gzip_batch(plain, strategy=1))
    got, got_off, status = check(codec, batches)
    assert [int(s) for s in status[:7]] == [hip.NONE, C.OK, C.OK, C.OK, C.OK, hip.NONE, C.OK]
    assert all(int(s) == hip.INFLATE for s in status[7:-1]), [n for n, s in zip(names, status[7:]) if int(s) != hip.INFLATE]
    assert int(status[-1]) == C.OK


def test_damaged_members(codec):
    """A thousand members with a few bytes changed, cut short or lengthened: whatever the restatement makes of
    each, the device makes the same."""
    rng = np.random.default_rng(78)
    words = [b"alpha", b"beta", b"gamma", b"delta"]
    records = [C.Record(0, 0, i, i, b"k%d" % (i % 7), b" ".join(words[int(x)] for x in rng.integers(0, 4, 6))) for i in range(60)]
    plain = build_batch(dict(base_offset=5, last_offset_delta=59), records)
    zeros = build_batch(dict(base_offset=6), [C.Record(0, 0, 0, 0, None, bytes(3000))])
    sources = [gzip_batch(plain), gzip_batch(plain, level=0), gzip_batch(plain, strategy=4), gzip_batch(zeros), gzip_batch(plain, chunk=500),
               gzip_batch(plain, name=b"n", extra=b"xy")]
    batches = []
    for i in range(1000):
        b = bytearray(sources[i % len(sources)])
        how = i % 5
        if how < 3:
            for at in rng.integers(61, len(b), 1 + how):
                b[at] = int(rng.integers(0, 256))
        elif how == 3:
            cut = int(rng.integers(62, len(b)))
            b = b[:cut]
            struct.pack_into(">i", b, 8, len(b) - 12)
        else:
            b += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
            struct.pack_into(">i", b, 8, len(b) - 12)
        batches.append(bytes(b))
    got, got_off, status = check(codec, batches)
    kinds = {int(s) for s in status}
    assert hip.INFLATE in kinds and C.OK in kinds


def test_workload_of_many_batches(codec):
    rng = np.random.default_rng(14)
    words = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta"]
    plain = []
    for b in range(16):
        records = []
        for i in range(300 + 17 * b):
            value = b" ".join(words[int(x)] for x in rng.integers(0, 8, 10))
            headers = [C.Header(b"trace", bytes(rng.integers(0, 256, 8, dtype=np.uint8))), C.Header(b"kind", words[i % 8])]
            records.append(C.Record(0, 0, i, i, b"user-%06d" % int(rng.integers(0, 5000)), value, headers))
        plain.append(build_batch(dict(base_offset=1000 * b, last_offset_delta=len(records) - 1), records))
    batches = [gzip_batch(p, **OPTIONS[i % len(OPTIONS)]) for i, p in enumerate(plain)]
    got, got_off, status = check(codec, batches)
    assert all(int(s) == C.OK for s in status)
    cols, totals, _, _ = gpu_decode(codec, got, got_off)
    want, want_totals, _, _ = gpu_decode(codec, b"".join(plain), offsets_of(plain))
    assert (totals.n_records, totals.n_headers, totals.n_failed) == (want_totals.n_records, want_totals.n_headers, 0)
    g, w = cols.to_host(), want.to_host()
    header_level = {f for _, fields in C.HEADER_GROUPS for f in fields}
    for name in g:
        if name not in ("crc",):
            n = totals.n_headers if name in header_level else len(g[name])
            assert (g[name][:n] == w[name][:n]).all(), name


def test_a_frame_of_thousands_of_small_batches(codec):
    """More batches than one pass of the layout's scan takes (1024), all four kinds and plain ones mixed, a
    damaged one every so often: offsets, statuses and bytes against the restatement."""
    rng = np.random.default_rng(99)
    batches = []
    for i in range(3000):
        value = bytes(rng.integers(97, 100, int(rng.integers(1, 120)), dtype=np.uint8))
        plain = build_batch(dict(base_offset=i), [C.Record(0, 0, 0, 0, b"k%d" % i, value)])
        kind = i % 5
        b = [plain, gzip_batch(plain), snappy_batch(plain), lz4_batch(plain), gzip_batch(plain, level=0)][kind]
        if i % 97 == 0 and kind:
            b = bytearray(b)
            b[-3] ^= 0x5A
            b = bytes(b)
        batches.append(b)
    got, got_off, status = check(codec, batches)
    assert {int(s) for s in status} == {C.OK, hip.NONE, hip.INFLATE}
