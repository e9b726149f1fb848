"""The LZ4, gzip and zstd deflate (into_record_data's three arms, nisshi-sans-io/src/record/deflated.rs:341-386,
as deflated::Batch::try_from(inflated::Batch) calls it, deflated.rs:392-409) through the C ABI.  The reference
holds no vector of compressed output and a compressor's bytes are its own choice, so parity here is the round
trip: what the device compressed must inflate -- by the CPU checker's restatement of the frame format, and by the
device's own inflate -- to the batch it was made from, with the header a compressed batch has (batch_length,
codec bits, a CRC that the checker's CRC-32C confirms); and it must have got smaller where there was
anything to find; a gzip member must also be one to Python's zlib (gzip.decompress checks header, blocks, CRC-32
and length) and a zstd frame one to libzstd (pyarrow's decompress).  **Parity of the compressed bytes themselves is unpinned** (the reference has no fixture)."""
import gzip
import struct

import numpy as np
import pytest

import oracle
from tansu_amd import columns as C, hip, synth
from util import build_batch, offsets_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def split(frame: bytes, off):
    return [frame[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]


def check_round_trip(codec, batches, codecs):
    """Deflate on the device (both entry points); every compressed batch against the checker."""
    import torch
    blob = b"".join(batches)
    off = offsets_of(batches)
    got, got_off, status, totals = codec.deflate(blob, off, codecs)
    assert len(got) == int(got_off[-1]) == totals.out_bytes
    out = split(got, got_off)
    for b, (was, now, want, st) in enumerate(zip(batches, out, codecs, status)):
        if want in (1, 3, 4) and st == C.OK:
            assert now[:8] == was[:8] and now[12:17] == was[12:17]                       # base offset, epoch, magic
            assert struct.unpack_from(">i", now, 8)[0] == len(now) - 12                  # batch_length
            assert struct.unpack_from(">h", now, 21)[0] == (struct.unpack_from(">h", was, 21)[0] | want)
            assert now[23:61] == was[23:61]                                              # the rest of the header
            assert struct.unpack_from(">I", now, 17)[0] == oracle.crc32c(now[21:])
            st_i, data = {3: oracle.lz4_inflate, 1: oracle.gzip_inflate, 4: oracle.zstd_inflate}[want](now[61:])
            assert st_i == C.OK and data == was[61:], b
            if want == 1:
                # a gzip member as Python's zlib reads it: header, blocks, CRC-32 and length of the trailer
                assert gzip.decompress(now[61:]) == was[61:], b
            if want == 4 and len(was) > 61:
                # a zstd frame as libzstd reads it (pyarrow's)
                import pyarrow as pa
                assert pa.decompress(now[61:], decompressed_size=len(was) - 61, codec="zstd", asbytes=True) == was[61:], b
        else:
            assert now == was, b
    # the device's own inflate gives the frame back (but for the crc field, which stays the compressed batch's)
    if any(c in (1, 3, 4) for c in codecs):
        back, back_off, back_status, _ = codec.inflate(got, got_off)
        for b, (was, now) in enumerate(zip(batches, split(back, back_off))):
            assert now[:17] == was[:17] and now[21:] == was[21:], b
    # the device form
    d_blob, d_off = codec.to_device(np.frombuffer(blob, dtype=np.uint8), off)
    d_codec = torch.from_numpy(np.asarray(codecs, dtype=np.uint8)).to(codec.device)
    d_out, d_out_off, d_status, t2 = codec.deflate_dev(d_blob, len(blob), d_off, len(batches), d_codec)
    assert d_out[:t2.out_bytes].cpu().numpy().tobytes() == got
    assert (d_status.cpu().numpy()[:len(batches)] == status).all()
    return out, status, totals


def records_batch(n, seed, words=True):
    rng = np.random.default_rng(seed)
    vocab = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta"]
    records = []
    for i in range(n):
        if words:
            value = b" ".join(vocab[int(x)] for x in rng.integers(0, 8, 10))
        else:
            value = bytes(rng.integers(0, 256, int(rng.integers(10, 200)), dtype=np.uint8))
        headers = [C.Header(b"trace", bytes(rng.integers(0, 256, 8, dtype=np.uint8)))] if i % 2 else []
        records.append(C.Record(0, 0, i, i, b"user-%06d" % int(rng.integers(0, 5000)), value, headers))
    return build_batch(dict(base_offset=1000 * seed, last_offset_delta=max(n - 1, 0)), records)


@pytest.mark.parametrize("want", [3, 1, 4])
def test_batches_of_records_get_smaller_and_come_back(codec, want):
    batches = [records_batch(n, 10 + i) for i, n in enumerate((1, 5, 40, 300, 1000, 2500))]
    out, status, totals = check_round_trip(codec, batches, [want] * len(batches))
    assert (status == C.OK).all() and totals.n_failed == 0
    assert sum(map(len, out[3:])) < 0.6 * sum(map(len, batches[3:]))


@pytest.mark.parametrize("want", [3, 1, 4])
def test_sizes_around_the_blocks_and_the_ends(codec, want):
    """Record data of 0 bytes (a frame with no block), of a few bytes (all literals: no match may start in the last
    twelve), around one, two and four blocks of 8 KiB, of long runs (matches of thousands of bytes, lengths around 15 and
    255 + 15), and of random bytes (stored blocks)."""
    rng = np.random.default_rng(3)
    datas = [b"", b"a", b"abcdefgh", bytes(12), bytes(13), bytes(19), b"ab" * 40]
    for n in (8191, 8192, 8193, 16383, 16384, 16385, 32768, 32769, 50000):
        datas.append((b"0123456789abcdef" * (n // 16 + 1))[:n])
        datas.append(bytes(rng.integers(0, 256, n, dtype=np.uint8)))
    datas.append(bytes(70000))
    datas.append(b"".join(bytes([i % 251]) * (i % 300 + 1) for i in range(400)))
    datas.append(b"".join(bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) + b"Z" * int(k2)
                          for k, k2 in zip(rng.integers(0, 40, 300), rng.integers(4, 300, 300))))
    # many short matches (four-byte words of a small vocabulary: a sequence every few bytes), and literal runs of
    # hundreds of bytes between matches (lengths whose codes take extra bits)
    words = [bytes(rng.integers(0, 256, 4, dtype=np.uint8)) for _ in range(40)]
    datas.append(b"".join(words[int(k)] for k in rng.integers(0, 40, 9000)))
    chunk = bytes(rng.integers(0, 256, 90, dtype=np.uint8))
    datas.append(b"".join(bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) + chunk for k in rng.integers(60, 700, 120)))
    head = records_batch(0, 1)[:61]
    batches = []
    for d in datas:
        h = bytearray(head)
        struct.pack_into(">i", h, 8, 49 + len(d))
        batches.append(bytes(h) + d)      # (the record data need not be records: the compressor does not look)
    out, status, totals = check_round_trip(codec, batches, [want] * len(batches))
    assert (status == C.OK).all()
    assert len(out[-5]) < 600            # 70 000 zero bytes
    assert len(out[-2]) < 0.8 * len(batches[-2]) and len(out[-1]) < len(batches[-1])
    if want == 1:
        # random bytes: stored blocks, five bytes a block and the member's eighteen
        assert len(out[8]) <= len(batches[8]) + 18 + 5 * 2


@pytest.mark.parametrize("seed", [41, 42])
def test_record_data_drawn_at_random(codec, seed):
    """Record data pieced together at random -- runs of one byte, random bytes, text of a small vocabulary,
    repeats of what came a drawn distance earlier -- of drawn lengths up to a few blocks, through the three
    codecs in one call."""
    rng = np.random.default_rng(seed)
    vocab = [b"offset", b"partition", b"topic", b"key", b"value", b"timestamp", b"\x00\x00\x00", b"{\"id\":", b"},", b"null"]
    head = records_batch(0, 1)[:61]
    batches, codecs = [], []
    for i in range(45):
        want = int(rng.integers(0, 30000)) if i % 5 else int(rng.integers(0, 40))
        d = bytearray()
        while len(d) < want:
            kind = int(rng.integers(0, 4))
            k = int(rng.integers(1, 600))
            if kind == 0:
                d += bytes([int(rng.integers(0, 256))]) * k
            elif kind == 1:
                d += bytes(rng.integers(0, 256, k, dtype=np.uint8))
            elif kind == 2:
                d += b" ".join(vocab[int(x)] for x in rng.integers(0, len(vocab), k // 6 + 1))
            elif d:
                back = int(rng.integers(1, min(len(d), 20000) + 1))
                for _ in range(k):
                    d.append(d[-back])
        d = bytes(d[:want])
        h = bytearray(head)
        struct.pack_into(">i", h, 8, 49 + len(d))
        batches.append(bytes(h) + d)
        codecs.append((3, 1, 4)[i % 3])
    out, status, totals = check_round_trip(codec, batches, codecs)
    assert (status == C.OK).all()
    assert sum(map(len, out)) < 0.8 * sum(map(len, batches))


def test_codecs_that_are_not_produced_and_headers_that_do_not_hold(codec):
    good = records_batch(200, 5)
    compressed = bytearray(good)
    compressed[22] |= 3                    # codec bits set already
    short = good[:40]
    wrong_length = bytearray(good)
    struct.pack_into(">i", wrong_length, 8, len(good))    # (12 too many)
    batches = [good, good, good, good, good, good, bytes(compressed), short, bytes(wrong_length), good, bytes(compressed)]
    codecs = [0, 1, 2, 4, 5, 3, 3, 3, 3, 3, 1]
    out, status, totals = check_round_trip(codec, batches, codecs)
    assert list(status) == [C.OK, C.OK, C.COMPRESSED, C.OK, C.UNKNOWN_COMPRESSION, C.OK,
                            C.MALFORMED, C.MALFORMED, C.MALFORMED, C.OK, C.MALFORMED]
    assert totals.n_failed == 6
    assert len(out[5]) < len(good) and out[9] == out[5] and len(out[1]) < len(good) and len(out[3]) < len(good)


@pytest.mark.parametrize("want", [3, 1, 4])
def test_encode_then_deflate_then_inflate_then_decode(codec, want):
    """The chain a fetch of a topic with a codec makes: columns -> encode -> deflate; and back: inflate -> decode
    gives the columns of the uncompressed frame."""
    import torch
    from test_gpu_parity import gpu_decode
    shape, counts, seed = synth.config("c2", 0.004)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    n = len(off) - 1
    cols, totals, d_blob, d_off = gpu_decode(codec, blob, off)
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, n, d_blob, cols.t["data_off"], out, len(blob) + 64, out_off)
    d_codec = torch.full((n,), want, dtype=torch.uint8, device=codec.device)
    d_packed, d_packed_off, d_status, t2 = codec.deflate_dev(out, t.out_bytes, out_off, n, d_codec)
    assert (d_status.cpu().numpy()[:n] == C.OK).all()
    d_back, d_back_off, d_st, t3 = codec.inflate_dev(d_packed, t2.out_bytes, d_packed_off, n)
    assert t3.out_bytes == t.out_bytes
    back = d_back[:t3.out_bytes].cpu().numpy()
    cols2, totals2, _, _ = gpu_decode(codec, back, d_back_off.cpu().numpy().astype(np.uint64))
    assert (totals2.n_records, totals2.n_headers, totals2.n_failed) == (nr, nh, 0)
    a, b = cols.to_host(nr, nh), cols2.to_host(nr, nh)
    for name in a:
        if name not in ("crc", "crc_computed", "enc_len"):
            assert (a[name] == b[name]).all(), name
