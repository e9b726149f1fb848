"""The CPU restatement of the LZ4 inflate (nisshi-sans-io/src/lib.rs:1752-1755: liblz4's frame decoder)
against the reference's vector (the codec test's batch, deflated.rs:762-850), XXH32's published answers, and
frames of every descriptor option and element form from a compressor of the tests' own."""
import struct

import numpy as np
import pytest

import oracle
from lz4_cases import corrupt_cases, frame, sequence
from tansu_amd import columns as C
from util import load_vectors, offsets_of, record_from_json

V = load_vectors()
INFLATE = 11


def test_xxh32_known_answers():
    assert oracle.xxh32(b"") == 0x02CC5D05
    assert oracle.xxh32(b"abc") == 0x32D153FF
    assert oracle.xxh32(b"Nobody inspects the spammish repetition") == 0xE2293B2F
    assert oracle.xxh32(b"", 1) == 0x0B2CB792


def test_codec_test_batch():
    case = [c for c in V["compressed_batches"] if c["name"] == "decode_lz4"][0]
    batch = bytes(case["batch"])
    assert struct.unpack_from(">h", batch, 21)[0] & 7 == 3
    st, out, out_off, status = oracle.inflate_frame(batch, offsets_of([batch]))
    assert st == C.OK and int(status[0]) == C.OK
    st, cols, nr, nh = oracle.decode_columns(out, out_off)
    assert st == C.OK and int(cols["status"][0]) == C.OK and int(cols["attributes"][0]) & 7 == 0
    assert C.records_of_batch(cols, 0, out) == [record_from_json(r) for r in case["records"]]


def data_of(kind, n, rng):
    if kind == "text":
        return bytes(rng.integers(97, 101, n, dtype=np.uint8))
    if kind == "random":
        return bytes(rng.integers(0, 256, n, dtype=np.uint8))
    if kind == "zeros":
        return bytes(n)
    unit = bytes(rng.integers(0, 256, min(n, 40000), dtype=np.uint8))     # repeats 40 000 bytes apart
    return (unit * (n // max(len(unit), 1) + 1))[:n] if n else b""


@pytest.mark.parametrize("kind", ["text", "random", "zeros", "far"])
def test_round_trip_of_generated_frames(kind):
    rng = np.random.default_rng(len(kind))
    for n in (0, 1, 12, 13, 19, 300, 5000, 70000, 200000):
        data = data_of(kind, n, rng)
        for opts in (dict(), dict(linked=True), dict(content_size=True, content_sum=True), dict(block_sums=True, size_id=5),
                     dict(store=True), dict(linked=True, block_bytes=3000), dict(block_bytes=1000, block_sums=True, content_sum=True)):
            st, got = oracle.lz4_inflate(frame(data, **opts) + b"what lies behind the frame")
            assert st == C.OK and got == data, (kind, n, opts)


def test_length_bytes():
    # literal and match lengths around the steps of 15 and 255
    for ll in (14, 15, 16, 269, 270, 271, 525):
        for ml in (4, 18, 19, 20, 273, 274, 275):
            lit = bytes(range(97, 123)) * 30
            body = sequence(lit[:ll] + b"wxyz", 4, ml) + sequence(b"123456789012")
            f = frame(b"")[:7] + struct.pack("<I", len(body)) + body + struct.pack("<I", 0)
            st, got = oracle.lz4_inflate(f)
            want = lit[:ll] + b"wxyz" + (b"wxyz" * (ml // 4 + 1))[:ml] + b"123456789012"
            assert st == C.OK and got == want, (ll, ml)


def test_frames_that_do_not_inflate():
    for name, f in corrupt_cases().items():
        st, got = oracle.lz4_inflate(f)
        assert st == INFLATE, name
