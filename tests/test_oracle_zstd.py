"""The CPU restatement of the zstd inflate (nisshi-sans-io/src/lib.rs:1761: the zstd crate's stream
Decoder, libzstd) against the reference's vector (the codec test's batch, deflated.rs:676-760), XXH64's
published answers, and frames from a real compressor (pyarrow's libzstd) at several levels: the restatement
must give back what the compressor was given."""
import struct

import numpy as np
import pytest

import oracle
from tansu_amd import columns as C
from util import load_vectors, offsets_of, record_from_json
from zstd_cases import (KINDS, LEVELS, SIZES, compress, corrupt_cases, data_of, skippable, stored_frame, with_checksum,
                        without_size)

V = load_vectors()
INFLATE = 11


def test_xxh64_known_answers():
    assert oracle.xxh64(b"") == 0xEF46DB3751D8E999
    assert oracle.xxh64(b"abc") == 0x44BC2CF5AD770999
    assert oracle.xxh64(b"Nobody inspects the spammish repetition") == 0xFBCEA83C8A378BF1
    assert oracle.xxh64(b"abc", 1) == 0xBEA9CA8199328908


def test_codec_test_batch():
    case = [c for c in V["compressed_batches"] if c["name"] == "decode_zstd"][0]
    batch = bytes(case["batch"])
    assert struct.unpack_from(">h", batch, 21)[0] & 7 == 4
    st, out, out_off, status = oracle.inflate_frame(batch, offsets_of([batch]))
    assert st == C.OK and int(status[0]) == C.OK
    st, cols, nr, nh = oracle.decode_columns(out, out_off)
    assert st == C.OK and int(cols["status"][0]) == C.OK and int(cols["attributes"][0]) & 7 == 0
    assert C.records_of_batch(cols, 0, out) == [record_from_json(r) for r in case["records"]]


@pytest.mark.parametrize("kind", KINDS)
def test_round_trip_of_compressed_frames(kind):
    rng = np.random.default_rng(len(kind))
    for n in SIZES:
        data = data_of(kind, n, rng)
        for level in LEVELS:
            frame = compress(data, level)
            st, got = oracle.zstd_inflate(frame)
            assert st == C.OK and got == data, (kind, n, level)


def test_frames_the_compressor_does_not_write():
    rng = np.random.default_rng(5)
    text = data_of("text", 150000, rng)
    frame = compress(text, 3)
    for made, want in ((with_checksum(frame, text), text), (without_size(frame), text),
                       (with_checksum(without_size(frame), text), text),
                       (stored_frame(text[:5000]), text[:5000]), (stored_frame(b""), b""),
                       (stored_frame(bytes(3000), rle_byte=0), bytes(3000)),
                       (stored_frame(b"z" * 2500, block=999, rle_byte=122), b"z" * 2500),
                       (skippable(b"not for the decoder") + frame, text), (frame + skippable(b"", 7), text),
                       (frame + compress(text[:777], 1) + stored_frame(b"tail"), text + text[:777] + b"tail"),
                       (b"", b"")):
        st, got = oracle.zstd_inflate(made)
        assert st == C.OK and got == want


def test_damaged_frames_do_not_inflate():
    rng = np.random.default_rng(6)
    text = data_of("text", 20000, rng)
    flipped_but_sound = 0
    for name, data in corrupt_cases(rng):
        st, got = oracle.zstd_inflate(data)
        if name.startswith("bit flipped") and st == C.OK:
            # (a changed bit of a literal or of an extra-bits field gives other bytes of the same length: the
            # frame has no checksum to tell)
            assert len(got) == len(text)
            flipped_but_sound += 1
            continue
        assert st == INFLATE, name
    assert flipped_but_sound < 30
