"""Parity of the LZ4 inflate (nisshi-sans-io/src/lib.rs:1752-1755 in Vec<Record>::try_from,
nisshi-sans-io/src/record/deflated.rs:444-457) with its CPU restatement, through the C ABI:
byte-exact output frames, statuses, and the records a decode of the output gives."""
import struct

import numpy as np
import pytest

import oracle
import lz4_cases
import snappy_cases
from tansu_amd import columns as C, hip
from test_gpu_inflate import check, snappy_batch
from test_gpu_parity import gpu_decode
from test_oracle_lz4 import data_of
from util import build_batch, load_vectors, offsets_of, record_from_json

pytestmark = pytest.mark.gpu
V = load_vectors()


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def lz4_batch(batch: bytes, **opts) -> bytes:
    """An uncompressed batch as the LZ4 batch of the same records (the CRC recomputed)."""
    return framed_batch(batch, lz4_cases.frame(batch[61:], **opts))


def framed_batch(batch: bytes, data: bytes) -> bytes:
    head = bytearray(batch[:61])
    struct.pack_into(">i", head, 8, 49 + len(data))
    struct.pack_into(">h", head, 21, (struct.unpack_from(">h", head, 21)[0] & ~7) | 3)
    struct.pack_into(">I", head, 17, oracle.crc32c(bytes(head[21:]) + data))
    return bytes(head) + data


OPTIONS = [dict(), dict(linked=True), dict(content_size=True, content_sum=True), dict(block_sums=True, size_id=5),
           dict(store=True), dict(linked=True, block_bytes=3000), dict(block_bytes=1000, block_sums=True, content_sum=True),
           dict(linked=True, content_sum=True, size_id=7)]


def test_reference_vector(codec):
    case = [c for c in V["compressed_batches"] if c["name"] == "decode_lz4"][0]
    got, got_off, status = check(codec, [bytes(case["batch"])])
    assert int(status[0]) == C.OK
    cols, totals, _, _ = gpu_decode(codec, got, got_off)
    assert C.records_of_batch(cols.to_host(), 0, got) == [record_from_json(r) for r in case["records"]]


@pytest.mark.parametrize("kind", ["text", "random", "zeros", "far"])
def test_generated_batches(codec, kind):
    rng = np.random.default_rng(len(kind) + 40)
    batches, plains = [], []
    for n in (1, 12, 13, 19, 300, 5000, 70000, 200000):
        value = data_of(kind, n, rng)
        plain = build_batch(dict(base_offset=n), [C.Record(0, 0, 0, 0, b"k", value), C.Record(0, 0, 1, 1, None, value[:9])])
        for opts in OPTIONS[(n % 3)::3] + [OPTIONS[n % len(OPTIONS)]]:
            batches.append(lz4_batch(plain, **opts))
            plains.append(plain)
    got, got_off, status = check(codec, batches)
    assert all(int(s) == C.OK for s in status)
    for i, plain in enumerate(plains):
        a = got[int(got_off[i]):int(got_off[i + 1])]
        assert a[:17] == plain[:17] and a[21:] == plain[21:]


def test_length_bytes_and_overlaps(codec):
    batches = []
    plain = build_batch(dict(base_offset=1), [C.Record(0, 0, 0, 0, None, b"v")])
    lit = bytes(range(97, 123)) * 30
    for ll in (14, 15, 16, 269, 270, 271, 525):
        for ml, off in ((4, 4), (18, 1), (19, 2), (20, 3), (273, 4), (274, 5), (275, 4), (1000, 63), (1000, 64), (1000, 65)):
            off = min(off, ll + 4)                 # (no further back than what the literals wrote)
            body = lz4_cases.sequence(lit[:ll] + b"wxyz", off, ml) + lz4_cases.sequence(b"123456789012")
            f = lz4_cases.frame(b"")[:7] + struct.pack("<I", len(body)) + body + struct.pack("<I", 0)
            batches.append(framed_batch(plain, f))
    got, got_off, status = check(codec, batches)
    assert all(int(s) == C.OK for s in status)


def test_frames_that_do_not_inflate_and_a_mixed_frame(codec):
    plain = build_batch(dict(base_offset=3), [C.Record(0, 0, i, i, b"key%d" % i, b"value" * i) for i in range(40)])
    zs = bytes([c for c in V["compressed_batches"] if c["name"] == "decode_zstd"][0]["batch"])
    batches = [plain, lz4_batch(plain), zs, snappy_batch(plain), plain[:30], lz4_batch(plain, linked=True, content_sum=True)]
    names = []
    for name, f in lz4_cases.corrupt_cases().items():
        batches.append(framed_batch(plain, f))
        names.append(name)
    # a content checksum that the bytes do not meet: found only once the bytes are there
    f = bytearray(lz4_cases.frame(plain[61:], content_sum=True))
    f[-1] ^= 0x40
    batches.append(framed_batch(plain, bytes(f)))
    batches.append(lz4_batch(plain, block_sums=True))
    got, got_off, status = check(codec, batches)
    assert [int(s) for s in status[:6]] == [hip.NONE, C.OK, C.OK, C.OK, hip.NONE, C.OK]
    assert all(int(s) == hip.INFLATE for s in status[6:-1]), [n for n, s in zip(names, status[6:]) if int(s) != hip.INFLATE]
    assert int(status[-1]) == C.OK


def test_workload_of_many_batches(codec):
    rng = np.random.default_rng(13)
    words = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta"]
    plain = []
    for b in range(16):
        records = []
        for i in range(300 + 17 * b):
            value = b" ".join(words[int(x)] for x in rng.integers(0, 8, 10))
            headers = [C.Header(b"trace", bytes(rng.integers(0, 256, 8, dtype=np.uint8))), C.Header(b"kind", words[i % 8])]
            records.append(C.Record(0, 0, i, i, b"user-%06d" % int(rng.integers(0, 5000)), value, headers))
        plain.append(build_batch(dict(base_offset=1000 * b, last_offset_delta=len(records) - 1), records))
    batches = [lz4_batch(p, **OPTIONS[i % len(OPTIONS)]) for i, p in enumerate(plain)]
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


def test_small_batches_of_large_blocks_ask_for_little_room(codec):
    """Frames that announce 4 MiB blocks and hold a few hundred bytes: the room the call asks for follows the bytes
    (255 a byte of a block at most), not the announced block size."""
    plains = [build_batch(dict(base_offset=i), [C.Record(0, 0, 0, 0, b"k%d" % i, b"value " * (i % 50 + 1))]) for i in range(300)]
    batches = [lz4_batch(p, size_id=7, linked=bool(i & 1)) for i, p in enumerate(plains)]
    blob = b"".join(batches)
    totals = hip.Totals()
    arr = np.frombuffer(blob, dtype=np.uint8)
    off = offsets_of(batches)
    out_off, status = np.zeros(len(batches) + 1, dtype=np.uint64), np.zeros(len(batches), dtype=np.int32)
    import ctypes
    st = codec.lib.nisshi_hip_inflate(codec.ctx, arr.ctypes.data, len(arr), off.ctypes.data, len(batches), None, 0,
                                      out_off.ctypes.data, status.ctypes.data, ctypes.byref(totals))
    assert st == C.CAPACITY
    assert len(b"".join(plains)) <= totals.out_bytes <= 255 * len(blob)
    got, got_off, status = check(codec, batches)
    assert all(int(s) == C.OK for s in status)
    for i, plain in enumerate(plains):
        a = got[int(got_off[i]):int(got_off[i + 1])]
        assert a[:17] == plain[:17] and a[21:] == plain[21:]


def test_damaged_streams_of_both_codecs(codec):
    """A thousand Snappy blocks and LZ4 frames with a few bytes changed, cut short or lengthened: whatever the
    restatement makes of each (most do not inflate, some inflate to other bytes), the device makes the same."""
    rng = np.random.default_rng(77)
    words = [b"alpha", b"beta", b"gamma", b"delta"]
    records = [C.Record(0, 0, i, i, b"k%d" % (i % 7), b" ".join(words[int(x)] for x in rng.integers(0, 4, 6))) for i in range(60)]
    plain = build_batch(dict(base_offset=5, last_offset_delta=59), records)
    zeros = build_batch(dict(base_offset=6), [C.Record(0, 0, 0, 0, None, bytes(3000))])
    sources = [snappy_batch(plain), snappy_batch(plain, xerial=True), snappy_batch(zeros), lz4_batch(plain), lz4_batch(plain, linked=True, block_bytes=700),
               lz4_batch(plain, content_size=True, content_sum=True, block_sums=True), lz4_batch(zeros), lz4_batch(plain, store=True)]
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
