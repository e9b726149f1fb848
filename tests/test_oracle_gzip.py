"""The CPU restatement of the gzip inflate (nisshi-sans-io/src/lib.rs:1716: flate2's GzDecoder) against the
reference's vector (the codec test's batch, deflated.rs:589-674) and against Python's zlib, a real DEFLATE
compressor: members of every block kind and header field must inflate to what zlib was given."""
import struct
import zlib

import numpy as np
import pytest

import oracle
from gzip_cases import corrupt_cases, member
from tansu_amd import columns as C
from util import load_vectors, offsets_of, record_from_json

V = load_vectors()
INFLATE = 11


def test_codec_test_batch():
    case = [c for c in V["compressed_batches"] if c["name"] == "decode_gzip"][0]
    batch = bytes(case["batch"])
    assert struct.unpack_from(">h", batch, 21)[0] & 7 == 1
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
    words = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta", b"user-", b"trace"]
    return b" ".join(words[int(x)] for x in rng.integers(0, 10, n // 5 + 1))[:n]


OPTIONS = [dict(), dict(level=0), dict(level=1), dict(level=9), dict(strategy=zlib.Z_FIXED), dict(strategy=zlib.Z_HUFFMAN_ONLY),
           dict(strategy=zlib.Z_RLE), dict(chunk=1000), dict(extra=b"ab\0cd", name=b"records", comment=b"of a batch"), dict(name=b"")]


@pytest.mark.parametrize("kind", ["text", "random", "zeros", "words"])
def test_members_from_zlib(kind):
    rng = np.random.default_rng(len(kind) + 5)
    for n in (0, 1, 5, 300, 5000, 70000, 200000):
        data = data_of(kind, n, rng)
        for opts in OPTIONS:
            st, got = oracle.gzip_inflate(member(data, **opts))
            assert st == C.OK and got == data, (kind, n, opts)


def test_members_that_do_not_inflate():
    for name, m in corrupt_cases().items():
        st, got = oracle.gzip_inflate(m)
        assert st == INFLATE, name
