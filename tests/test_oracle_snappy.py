"""The CPU restatement of the Snappy inflate (nisshi-sans-io/src/lib.rs:1717-1751, the `snap` crate's
raw decoder) against the reference's own vectors: the codec test's batch (deflated.rs:852-905), the two
compressed produce requests of tests/decode.rs with their fully expanded inflated batches, and the three
producer frames of tests/snappy.rs with what they assert of their one record."""
import struct

import numpy as np
import pytest

import oracle
from snappy_cases import compress, corrupt_cases
from tansu_amd import columns as C
from util import find_batches, load_vectors, offsets_of, record_from_json

V = load_vectors()
INFLATE = 11


def inflated_records(batch: bytes):
    """Records of one Snappy batch through the oracle: inflate the frame of one batch, decode it."""
    st, out, out_off, status = oracle.inflate_frame(batch, offsets_of([batch]))
    assert st == C.OK and int(status[0]) == C.OK
    st, cols, nr, nh = oracle.decode_columns(out, out_off)
    assert st == C.OK and int(cols["status"][0]) == C.OK
    return cols, C.records_of_batch(cols, 0, out), out


def test_codec_test_batch():
    case = [c for c in V["compressed_batches"] if c["name"] == "decode_snappy"][0]
    batch = bytes(case["batch"])
    assert struct.unpack_from(">h", batch, 21)[0] & 7 == 2
    cols, records, out = inflated_records(batch)
    assert records == [record_from_json(r) for r in case["records"]]
    assert int(cols["attributes"][0]) & 7 == 0
    # everything else of the header is the compressed batch's
    assert out[:8] == batch[:8] and out[12:21] == batch[12:21] and out[23:61] == batch[23:61]


@pytest.mark.parametrize("case", V["compressed_frames"], ids=lambda c: c["name"])
def test_compressed_produce_requests(case):
    frame = bytes(case["frame"])
    (a, b), = find_batches(frame)
    batch = frame[a:b]
    codec = struct.unpack_from(">h", batch, 21)[0] & 7
    want = case["batches"][0]
    if codec != 2:
        pytest.skip(f"codec {codec}: not Snappy")
    cols, records, out = inflated_records(batch)
    assert records == [record_from_json(r) for r in want["records"]]
    for k, w in want.items():
        # the inflated batch carries the compressed batch's batch_length (inflated.rs:114-150); the
        # rewritten wire batch must carry its own
        if k not in ("records", "batch_length"):
            assert int(cols[k][0]) == w, k
    assert want["batch_length"] == struct.unpack_from(">i", batch, 8)[0]


@pytest.mark.parametrize("case", V["snappy_frames"], ids=lambda c: c["name"])
def test_producer_frames(case):
    frame = bytes(case["frame"])
    (a, b), = find_batches(frame)
    batch = frame[a:b]
    assert batch[61:69] == b"\x82SNAPPY\x00"           # xerial framing: 20 bytes skipped
    if case["ignored"]:
        # the reference does not pass this one (#[ignore]): two xerial chunks, of which it knows one
        chunk = struct.unpack_from(">i", batch, 61 + 16)[0]
        assert 61 + 20 + chunk < len(batch)
        st, out, out_off, status = oracle.inflate_frame(batch, offsets_of([batch]))
        assert st == C.OK and int(status[0]) == INFLATE and out == batch
        return
    cols, records, out = inflated_records(batch)
    assert int(cols["record_count"][0]) == case["record_count"]
    assert len(records) == case["records"]
    assert records[0].length == case["first_record_length"]
    # the record's length is the size of its body: the inflated data is that record and nothing else
    assert len(out) == 61 + oracle.varint_size(records[0].length) + records[0].length


def test_round_trip_of_generated_streams():
    rng = np.random.default_rng(3)
    for n in (0, 1, 5, 60, 61, 300, 70000, 200000):
        for kind in ("text", "random", "zeros"):
            data = {"text": bytes(rng.integers(97, 101, n, dtype=np.uint8)),
                    "random": bytes(rng.integers(0, 256, n, dtype=np.uint8)), "zeros": bytes(n)}[kind]
            for xerial in (False, True):
                st, got = oracle.snappy_inflate(compress(data, xerial=xerial))
                assert st == C.OK and got == data, (n, kind, xerial)


def test_streams_that_do_not_inflate():
    for name, stream in corrupt_cases().items():
        st, got = oracle.snappy_inflate(stream)
        assert st == INFLATE, name


def test_frame_with_other_batches_passes_them_through():
    from util import build_batch
    plain = build_batch(dict(base_offset=3), [C.Record(0, 0, 0, 0, b"k", b"v")])
    case = [c for c in V["compressed_batches"] if c["name"] == "decode_snappy"][0]
    zs = bytes([c for c in V["compressed_batches"] if c["name"] == "decode_zstd"][0]["batch"])
    snappy = bytes(case["batch"])
    broken = bytearray(snappy)
    broken[61] += 1                                      # announces a byte more than the stream holds
    batches = [plain, snappy, zs, bytes(broken), plain[:30]]
    st, out, out_off, status = oracle.inflate_frame(b"".join(batches), offsets_of(batches))
    assert st == C.OK
    assert list(status) == [10, C.OK, C.OK, INFLATE, 10]     # (the zstd batch inflates too)
    pieces = [out[int(out_off[i]):int(out_off[i + 1])] for i in range(5)]
    assert pieces[0] == plain and pieces[3] == bytes(broken) and pieces[4] == plain[:30]
    assert len(pieces[2]) == 61 + 454 and pieces[2][61:] == pieces[1][61:]      # (the same record as the Snappy batch's)
    assert len(pieces[1]) == 61 + 454
