"""Parity of the zstd inflate (nisshi-sans-io/src/lib.rs:1761-1764 in Vec<Record>::try_from,
nisshi-sans-io/src/record/deflated.rs:444-457) with its CPU restatement, through the C ABI:
byte-exact output frames, statuses, and the records a decode of the output gives.  The frames
come from a real compressor (pyarrow's libzstd) and from variations of its frames."""
import struct

import numpy as np
import pytest

import oracle
import zstd_cases
from tansu_amd import columns as C, hip
from test_gpu_inflate import check, snappy_batch
from test_gpu_parity import gpu_decode
from util import build_batch, load_vectors, offsets_of, record_from_json

pytestmark = pytest.mark.gpu
V = load_vectors()


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def framed_batch(batch: bytes, data: bytes) -> bytes:
    """The batch's header with `data` as zstd record data (the CRC recomputed)."""
    head = bytearray(batch[:61])
    struct.pack_into(">i", head, 8, 49 + len(data))
    struct.pack_into(">h", head, 21, (struct.unpack_from(">h", head, 21)[0] & ~7) | 4)
    struct.pack_into(">I", head, 17, oracle.crc32c(bytes(head[21:]) + data))
    return bytes(head) + data


def zstd_batch(batch: bytes, level: int = 3, how: str = "plain") -> bytes:
    """An uncompressed batch as the zstd batch of the same records."""
    frame = zstd_cases.compress(batch[61:], level)
    if how == "checksum":
        frame = zstd_cases.with_checksum(frame, batch[61:])
    elif how == "streamed":
        frame = zstd_cases.without_size(frame)
    elif how == "streamed+checksum":
        frame = zstd_cases.with_checksum(zstd_cases.without_size(frame), batch[61:])
    elif how == "skippable":
        frame = zstd_cases.skippable(b"producer metadata") + frame + zstd_cases.skippable(b"", 3)
    return framed_batch(batch, frame)


HOW = ["plain", "checksum", "streamed", "streamed+checksum", "skippable"]


def test_reference_vector(codec):
    case = [c for c in V["compressed_batches"] if c["name"] == "decode_zstd"][0]
    got, got_off, status = check(codec, [bytes(case["batch"])])
    assert int(status[0]) == C.OK
    cols, totals, _, _ = gpu_decode(codec, got, got_off)
    assert C.records_of_batch(cols.to_host(), 0, got) == [record_from_json(r) for r in case["records"]]


@pytest.mark.parametrize("kind", zstd_cases.KINDS)
def test_generated_batches(codec, kind):
    rng = np.random.default_rng(len(kind) + 60)
    batches, plains = [], []
    for i, n in enumerate((1, 7, 100, 1000, 5000, 70000, 140000, 300000)):
        value = zstd_cases.data_of(kind, n, rng)
        plain = build_batch(dict(base_offset=n), [C.Record(0, 0, 0, 0, b"k", value), C.Record(0, 0, 1, 1, None, value[:9])])
        for level in zstd_cases.LEVELS:
            batches.append(zstd_batch(plain, level, HOW[(i + level) % len(HOW)]))
            plains.append(plain)
    got, got_off, status = check(codec, batches)
    assert all(int(s) == C.OK for s in status)
    for i, plain in enumerate(plains):
        a = got[int(got_off[i]):int(got_off[i + 1])]
        assert a[:17] == plain[:17] and a[21:] == plain[21:]


def test_frames_the_compressor_does_not_write(codec):
    plain = build_batch(dict(base_offset=3), [C.Record(0, 0, i, i, b"key%d" % i, b"value" * i) for i in range(40)])
    data = plain[61:]
    frames = [zstd_cases.stored_frame(data), zstd_cases.stored_frame(data, block=100),
              zstd_cases.compress(data[:500], 1) + zstd_cases.stored_frame(data[500:]),
              zstd_cases.skippable(b"x" * 300) + zstd_cases.compress(data, 19)]
    batches = [framed_batch(plain, f) for f in frames]
    zeros = build_batch(dict(base_offset=4), [C.Record(0, 0, 0, 0, None, bytes(2990))])
    # (a batch of zeros is not all zeros: the frame of repeated-byte blocks is checked by the restatement alone)
    batches.append(framed_batch(zeros, zstd_cases.compress(zeros[61:], 3)))
    got, got_off, status = check(codec, batches)
    assert all(int(s) == C.OK for s in status)
    for i in range(4):
        a = got[int(got_off[i]):int(got_off[i + 1])]
        assert a[21:] == plain[21:]
    # blocks of one byte repeated, and an empty frame: not record data, but the bytes must be the restatement's
    odd = [framed_batch(plain, zstd_cases.stored_frame(bytes(3000), rle_byte=0)),
           framed_batch(plain, zstd_cases.stored_frame(b"z" * 2500, block=999, rle_byte=122)),
           framed_batch(plain, zstd_cases.stored_frame(b""))]
    got, got_off, status = check(codec, odd)
    assert all(int(s) == C.OK for s in status)


def test_frames_that_do_not_inflate_and_a_mixed_frame(codec):
    rng = np.random.default_rng(61)
    plain = build_batch(dict(base_offset=3), [C.Record(0, 0, i, i, b"key%d" % i, b"value" * i) for i in range(40)])
    batches = [plain, zstd_batch(plain), snappy_batch(plain), plain[:30], zstd_batch(plain, 19, "checksum")]
    names = []
    for name, data in zstd_cases.corrupt_cases(rng):
        batches.append(framed_batch(plain, data))
        names.append(name)
    got, got_off, status = check(codec, batches)     # (byte for byte and status for status what the restatement makes of them)
    assert [int(s) for s in status[:5]] == [hip.NONE, C.OK, C.OK, hip.NONE, C.OK]
    for name, s in zip(names, status[5:]):
        if not name.startswith("bit flipped"):
            assert int(s) == hip.INFLATE, name


def test_workload_of_many_batches(codec):
    rng = np.random.default_rng(14)
    words = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta"]
    plain = []
    for b in range(24):
        records = []
        for i in range(300 + 17 * b):
            value = b" ".join(words[int(x)] for x in rng.integers(0, 8, 10))
            headers = [C.Header(b"trace", bytes(rng.integers(0, 256, 8, dtype=np.uint8))), C.Header(b"kind", words[i % 8])]
            records.append(C.Record(0, 0, i, i, b"user-%06d" % int(rng.integers(0, 5000)), value, headers))
        plain.append(build_batch(dict(base_offset=1000 * b, last_offset_delta=len(records) - 1), records))
    batches = [zstd_batch(p, zstd_cases.LEVELS[i % 4], HOW[i % len(HOW)]) for i, p in enumerate(plain)]
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


def test_damaged_frames(codec):
    """Several hundred frames with a few bytes changed, cut short or lengthened: whatever the restatement
    makes of each (most do not inflate, some inflate to other bytes), the device makes the same."""
    rng = np.random.default_rng(79)
    words = [b"alpha", b"beta", b"gamma", b"delta"]
    records = [C.Record(0, 0, i, i, b"k%d" % (i % 7), b" ".join(words[int(x)] for x in rng.integers(0, 4, 6))) for i in range(200)]
    plain = build_batch(dict(base_offset=5, last_offset_delta=199), records)
    zeros = build_batch(dict(base_offset=6), [C.Record(0, 0, 0, 0, None, bytes(3000))])
    sources = [zstd_batch(plain, 1), zstd_batch(plain, 19, "checksum"), zstd_batch(zeros), zstd_batch(plain, 3, "streamed"),
               zstd_batch(plain, 9, "skippable")]
    batches = []
    for i in range(600):
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
