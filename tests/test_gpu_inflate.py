"""Parity of the Snappy inflate (nisshi-sans-io/src/lib.rs:1717-1751 in Vec<Record>::try_from,
nisshi-sans-io/src/record/deflated.rs:444-457) with its CPU restatement, through the C ABI:
byte-exact output frames, statuses, and the records a decode of the output gives."""
import struct

import numpy as np
import pytest

import oracle
from snappy_cases import compress, corrupt_cases
from tansu_amd import columns as C, hip, synth
from test_gpu_parity import gpu_decode
from util import build_batch, find_batches, load_vectors, offsets_of, record_from_json

pytestmark = pytest.mark.gpu
V = load_vectors()


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def snappy_batch(batch: bytes, xerial=False) -> bytes:
    """An uncompressed batch as the Snappy batch of the same records (the CRC recomputed)."""
    data = compress(batch[61:], xerial=xerial)
    head = bytearray(batch[:61])
    struct.pack_into(">i", head, 8, 49 + len(data))
    struct.pack_into(">h", head, 21, (struct.unpack_from(">h", head, 21)[0] & ~7) | 2)
    body = bytes(head[21:]) + data
    struct.pack_into(">I", head, 17, oracle.crc32c(body))
    return bytes(head) + data


def check(codec, batches):
    blob, off = b"".join(batches), offsets_of(batches)
    st, want, want_off, want_status = oracle.inflate_frame(blob, off)
    assert st == C.OK
    got, got_off, got_status, totals = codec.inflate(blob, off)
    assert list(got_status) == list(want_status)
    assert list(got_off) == list(want_off)
    assert got == want
    assert totals.out_bytes == len(want) and totals.n_failed == int((want_status == hip.INFLATE).sum())
    # the device form gives the same
    import torch
    d_blob, d_off = codec.to_device(np.frombuffer(blob, dtype=np.uint8), off)
    d_out, d_out_off, d_status, t2 = codec.inflate_dev(d_blob, len(blob), d_off, len(batches))
    assert d_out[:t2.out_bytes].cpu().numpy().tobytes() == want
    assert d_out_off.cpu().numpy().tolist() == [int(x) for x in want_off]
    return got, got_off, got_status


def test_reference_vectors(codec):
    batches = [bytes([c for c in V["compressed_batches"] if c["name"] == "decode_snappy"][0]["batch"])]
    for case in V["compressed_frames"] + V["snappy_frames"]:
        frame = bytes(case["frame"])
        (a, b), = find_batches(frame)
        batches.append(frame[a:b])
    got, got_off, status = check(codec, batches)
    # the records of the codec test's batch, as the reference expands them
    cols, totals, _, _ = gpu_decode(codec, got, got_off)
    g = cols.to_host()
    want = [c for c in V["compressed_batches"] if c["name"] == "decode_snappy"][0]["records"]
    assert C.records_of_batch(g, 0, got) == [record_from_json(r) for r in want]
    ignored = [c["ignored"] for c in V["snappy_frames"]]
    assert [int(s) for s in status[-3:]] == [hip.INFLATE if i else C.OK for i in ignored]
    for k, case in enumerate(V["snappy_frames"]):
        b = len(batches) - 3 + k
        if not case["ignored"]:
            assert int(g["status"][b]) == C.OK and int(g["record_count"][b]) == case["record_count"]
            assert int(g["length"][int(g["rec_base"][b])]) == case["first_record_length"]


def test_mixed_frame_and_streams_that_do_not_inflate(codec):
    plain = build_batch(dict(base_offset=3), [C.Record(0, 0, 0, 0, b"k", b"v")])
    zs = bytes([c for c in V["compressed_batches"] if c["name"] == "decode_zstd"][0]["batch"])
    good = snappy_batch(build_batch(dict(base_offset=9), [C.Record(0, 0, i, i, b"key%d" % i, b"value" * i) for i in range(40)]))
    batches = [plain, good, zs, plain[:30], snappy_batch(plain, xerial=True)]
    for name, stream in corrupt_cases().items():
        head = bytearray(good[:61])
        struct.pack_into(">i", head, 8, 49 + len(stream))
        batches.append(bytes(head) + stream)
    got, got_off, status = check(codec, batches)
    assert [int(s) for s in status[:5]] == [hip.NONE, C.OK, C.OK, hip.NONE, C.OK]   # (the zstd batch inflates too)
    assert all(int(s) == hip.INFLATE for s in status[5:])


@pytest.mark.parametrize("kind", ["text", "random", "zeros", "far"])
def test_generated_batches(codec, kind):
    """Streams of every element kind: short and long literals, copies of all three forms, copies that
    overlap their own output, and copies from further back than the kernel keeps in LDS."""
    rng = np.random.default_rng(len(kind))
    batches = []
    for n in (1, 7, 60, 61, 300, 5000, 70000, 150000):
        if kind == "text":
            value = bytes(rng.integers(97, 101, n, dtype=np.uint8))
        elif kind == "random":
            value = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind == "zeros":
            value = bytes(n)
        else:
            unit = bytes(rng.integers(0, 256, min(n, 40000), dtype=np.uint8))   # repeats 40 000 bytes apart
            value = (unit * (n // len(unit) + 1))[:n]
        plain = build_batch(dict(base_offset=n), [C.Record(0, 0, 0, 0, b"k", value), C.Record(0, 0, 1, 1, None, value[:9])])
        batches.append(snappy_batch(plain, xerial=n % 2 == 0))
        batches.append(plain)
    got, got_off, status = check(codec, batches)
    assert all(int(s) in (C.OK, hip.NONE) for s in status)
    # what was compressed comes back as it was, but for the crc field, which stays the compressed batch's
    for i in range(0, len(batches), 2):
        a = got[int(got_off[i]):int(got_off[i + 1])]
        assert a[:17] == batches[i + 1][:17] and a[21:] == batches[i + 1][21:]


def test_workload_of_many_batches(codec):
    """Batches of hundreds of small records with keys, values and headers, compressed: the output against
    the oracle's, and the decode of the output against the decode of the batches before compression
    (every column but the CRC read from the header, which is the compressed batch's)."""
    rng = np.random.default_rng(12)
    words = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta"]
    plain = []
    for b in range(24):
        records = []
        for i in range(300 + 17 * b):
            value = b" ".join(words[int(x)] for x in rng.integers(0, 8, 10))
            headers = [C.Header(b"trace", bytes(rng.integers(0, 256, 8, dtype=np.uint8))), C.Header(b"kind", words[i % 8])]
            records.append(C.Record(0, 0, i, i, b"user-%06d" % int(rng.integers(0, 5000)), value, headers))
        plain.append(build_batch(dict(base_offset=1000 * b, last_offset_delta=len(records) - 1), records))
    batches = [snappy_batch(p, xerial=i % 3 == 0) for i, p in enumerate(plain)]
    assert sum(map(len, batches)) < 0.7 * sum(map(len, plain))
    got, got_off, status = check(codec, batches)
    assert all(int(s) == C.OK for s in status)
    cols, totals, _, _ = gpu_decode(codec, got, got_off)
    want, want_totals, _, _ = gpu_decode(codec, b"".join(plain), offsets_of(plain))
    assert (totals.n_records, totals.n_headers, totals.n_failed) == (want_totals.n_records, want_totals.n_headers, 0)
    g, w = cols.to_host(), want.to_host()
    header_level = {f for _, fields in C.HEADER_GROUPS for f in fields}
    for name in g:
        if name not in ("crc",):
            n = totals.n_headers if name in header_level else len(g[name])   # (slots past the headers are not written)
            assert (g[name][:n] == w[name][:n]).all(), name
    assert (g["crc_computed"] != g["crc"]).all()     # (the field is the compressed batch's)


def test_empty_call(codec):
    got, got_off, status, totals = codec.inflate(b"", np.zeros(1, dtype=np.uint64))
    assert got == b"" and list(got_off) == [0] and totals.out_bytes == 0
