"""Pin the CPU restatement (oracle/) against the golden vectors the reference's
own tests hold (tests/golden/vectors.json, made by tests/golden/extract_vectors.py)."""
import struct

import numpy as np
import pytest

import oracle
from tansu_amd import columns as C
from util import find_batches, load_vectors, offsets_of, record_from_json

V = load_vectors()


def test_crc32c_known_answers():
    assert oracle.crc32c(b"123456789") == 0xE3069283  # the standard CRC-32C check value
    kat = V["crc_known_answer"]
    assert oracle.crc32c(bytes(kat["bytes"])) == kat["crc"]
    assert oracle.crc32c(b"") == 0


def test_varint_vectors():
    for c in V["varints"]["signed"]:
        assert oracle.varint_encode(c["value"]) == bytes(c["bytes"])
        assert oracle.varint_decode(bytes(c["bytes"])) == (0, c["value"], len(c["bytes"]))
    for c in V["varints"]["unsigned"]:
        assert oracle.varint_encode(c["value"], "u32") == bytes(c["bytes"])


def test_varint_quirks():
    # up to 5 bytes; the 5th byte's bits above 32 are dropped; a 6th byte overflows
    assert oracle.varint_decode(bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x7F]))[0] == C.OK
    st, v, n = oracle.varint_decode(bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F]))
    assert (st, v, n) == (C.OK, -(2 ** 31), 5)
    st2, v2, _ = oracle.varint_decode(bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x7F]))
    assert v2 == v  # dropped bits
    assert oracle.varint_decode(bytes([0x80] * 5 + [0x00]))[0] == C.OVERFLOW
    assert oracle.varint_decode(bytes([0x80] * 5))[0] == C.TRY_GET  # under-run wins
    assert oracle.varint_decode(b"")[0] == C.TRY_GET
    # padded (non-canonical) encodings decode fine
    assert oracle.varint_decode(bytes([0x82, 0x00])) == (C.OK, 1, 2)
    # 64-bit: 10 bytes fine, 11th overflows
    assert oracle.varint_decode(bytes([0xFF] * 9 + [0x01]), "i64")[0] == C.OK
    assert oracle.varint_decode(bytes([0x80] * 10 + [0x00]), "i64")[0] == C.OVERFLOW
    for v in (0, 1, -1, 63, 64, -64, -65, 2 ** 31 - 1, -(2 ** 31)):
        e = oracle.varint_encode(v)
        assert len(e) == oracle.varint_size(v)
        assert oracle.varint_decode(e) == (C.OK, v, len(e))
    for v in (0, 1, -1, 2 ** 63 - 1, -(2 ** 63), 1 << 40):
        e = oracle.varint_encode(v, "i64")
        assert len(e) == oracle.varint_size(v, "i64")
        assert oracle.varint_decode(e, "i64") == (C.OK, v, len(e))


def test_octets_sizes():
    for c in V["octets_sizes"]["cases"]:
        n = -1 if c["octets"] is None else len(c["octets"])
        assert oracle.LIB.no_octets_size(n) == c["size"]
    for c in V["octets_sizes"]["sequences"]:
        size = oracle.varint_size(len(c["items"])) + sum(
            oracle.LIB.no_octets_size(len(i)) for i in c["items"])
        assert size == c["size"]


def test_attribute_bits_and_compression():
    import ctypes
    bits = V["attributes"]["bits"]
    for name, code in (("none", 0), ("gzip", 1), ("snappy", 2), ("lz4", 3), ("zstd", 4)):
        out = ctypes.c_int()
        assert oracle.LIB.no_compression(bits[name], ctypes.byref(out)) == C.OK
        assert out.value == code
    for code in V["bad_compression"]["codes"]:
        out = ctypes.c_int()
        assert oracle.LIB.no_compression(code, ctypes.byref(out)) == C.UNKNOWN_COMPRESSION
    # reset compression keeps the flag bits only
    out = ctypes.c_int16()
    attrs = bits["gzip"] | bits["log_append_time"] | bits["transaction"] | bits["control"] \
        | bits["delete_horizon"] | 0x100
    assert oracle.LIB.no_attributes_clear_compression(attrs, ctypes.byref(out)) == C.OK
    assert out.value == 8 | 16 | 32 | 64


def _decode_one(batch: bytes, by_value=True):
    st, cols, nr, nh = oracle.decode_columns(batch, offsets_of([batch]), by_value)
    assert st == C.OK
    return cols, nr, nh


def test_record_def_vector():
    v = V["record_def"]
    rec = record_from_json(v["record"])
    # wrap the record in a batch so the column decoder can be used
    hdr = dict(magic=2)
    cols, payload, src_off = C.columns_from_records([(hdr, [rec])])
    st, out, out_off = oracle.encode_columns(cols, 1, payload, src_off)
    assert st == C.OK
    assert out[61:] == bytes(v["bytes"])
    dcols, nr, nh = _decode_one(out)
    assert C.records_of_batch(dcols, 0, out) == [rec]
    # builder size: length excludes its own prefix (record.rs:465-475)
    st, bl, crc = oracle.builder_columns(cols, 1, payload, src_off)
    assert bl[0] == 45 + v["sequence_size"]


def test_batch_71():
    v = V["batch_71"]
    raw = bytes(v["bytes"])
    cols, nr, nh = _decode_one(raw)
    assert (nr, nh) == (1, 0)
    for k, want in v["batch"].items():
        if k != "records":
            assert int(cols[k][0]) == want, k
    assert int(cols["crc_computed"][0]) == v["batch"]["crc"]
    assert C.records_of_batch(cols, 0, raw) == [record_from_json(r) for r in v["batch"]["records"]]
    # re-encode is byte-identical
    st, out, _ = oracle.encode_columns(cols, 1, raw, cols["data_off"])
    assert st == C.OK and out == raw
    # Builder::build gives batch_length 59 and the same crc (inflated.rs:533-554)
    st, bl, crc = oracle.builder_columns(cols, 1, raw, cols["data_off"])
    assert (int(bl[0]), int(crc[0])) == (59, 1126819645)


@pytest.mark.parametrize("case", V["decode_frames"], ids=lambda c: c["name"])
def test_decode_frames(case):
    frame = bytes(case["frame"])
    spans = find_batches(frame)
    assert len(spans) == len(case["batches"])
    blob = b"".join(frame[a:b] for a, b in spans)
    st, off = oracle.frame_split(blob)
    assert st == C.OK and len(off) == len(spans) + 1
    st, cols, nr, nh = oracle.decode_columns(blob, off)
    assert st == C.OK
    for i, want in enumerate(case["batches"]):
        assert int(cols["status"][i]) == C.OK
        for k, w in want.items():
            if k != "records":
                assert int(cols[k][i]) == w, (i, k)
        assert int(cols["crc_computed"][i]) == want["crc"]
        assert C.records_of_batch(cols, i, blob) == [record_from_json(r) for r in want["records"]]
    # the reference's tests compare the decoded frame with the re-deflated
    # expected value, so the re-encode of every golden batch is pinned too
    st, out, out_off = oracle.encode_columns(cols, len(spans), blob, cols["data_off"])
    assert st == C.OK and out == blob


@pytest.mark.parametrize("case", V["encode_frames"] + V["bench_frames"], ids=lambda c: c["name"])
def test_wire_frames_round_trip(case):
    frame = bytes(case["frame"])
    spans = find_batches(frame)
    assert spans, "no batch found in frame"
    blob = b"".join(frame[a:b] for a, b in spans)
    st, off = oracle.frame_split(blob)
    assert st == C.OK
    st, cols, nr, nh = oracle.decode_columns(blob, off)
    assert st == C.OK and (cols["status"] == C.OK).all()
    assert (cols["crc"] == cols["crc_computed"]).all()
    st, out, _ = oracle.encode_columns(cols, len(spans), blob, cols["data_off"])
    assert st == C.OK and out == blob


def test_bench_frame_1k_values():
    case = [c for c in V["bench_frames"] if c["name"] == "produce_request_v11_001"][0]
    frame = bytes(case["frame"])
    (a, b), = find_batches(frame)
    cols, nr, nh = _decode_one(frame[a:b])
    assert nr == int(cols["record_count"][0]) == 7 and nh == 0
    assert (cols["val_len"] == 1024).all() and (cols["key_len"] == -1).all()
    assert list(cols["offset_delta"]) == list(range(7))


def test_control_batch():
    v = V["control_batch"]
    hdr = dict(v["batch"])
    data = bytes(v["record_data"])
    raw = struct.pack(">qiibIhiqqqhiI", hdr["base_offset"], hdr["batch_length"],
                      hdr["partition_leader_epoch"], hdr["magic"], hdr["crc"], hdr["attributes"],
                      hdr["last_offset_delta"], hdr["base_timestamp"], hdr["max_timestamp"],
                      hdr["producer_id"], hdr["producer_epoch"], hdr["base_sequence"],
                      hdr["record_count"]) + data
    assert hdr["batch_length"] == 49 + len(data)
    cols, nr, nh = _decode_one(raw)
    assert int(cols["attributes"][0]) & 16 and int(cols["attributes"][0]) & 32
    (rec,) = C.records_of_batch(cols, 0, raw)
    assert rec.key == bytes(v["key"]) and rec.value == bytes(v["value"])
    assert int(cols["crc_computed"][0]) == hdr["crc"]


def test_proptest_regression_round_trip():
    for case in V["proptest_regressions"]["cases"]:
        if case["compression"] != "None":
            continue
        recs = []
        for i, (k, v) in enumerate(zip(case["rec_keys"], case["rec_values"])):
            r = C.Record(0, 0, 0, i, None if k is None else bytes(k), None if v is None else bytes(v))
            recs.append(r)
        hdr = {k: case[k] for k in ("base_offset", "partition_leader_epoch", "base_timestamp",
                                    "max_timestamp", "producer_id", "producer_epoch",
                                    "base_sequence")}
        hdr["magic"] = 2
        cols, payload, src_off = C.columns_from_records([(hdr, recs)])
        st, out, out_off = oracle.encode_columns(cols, 1, payload, src_off)
        assert st == C.OK
        dcols, nr, nh = _decode_one(out)
        assert C.records_of_batch(dcols, 0, out) == recs
        st, again, _ = oracle.encode_columns(dcols, 1, out, dcols["data_off"])
        assert again == out


def test_batch_header_errors():
    raw = bytes(V["batch_71"]["bytes"])
    def status(buf):
        st, cols, _, _ = oracle.decode_columns(buf, offsets_of([buf]))
        assert st == C.OK
        return int(cols["status"][0])
    assert status(raw[:20]) == C.TRY_GET
    assert status(raw[:7]) == C.TRY_GET
    assert status(raw[:40]) == C.OVERFLOW           # crc_data_size > remaining
    neg = bytearray(raw); neg[8:12] = struct.pack(">i", -5)
    assert status(bytes(neg)) == C.TRY_FROM_INT
    small = bytearray(raw); small[8:12] = struct.pack(">i", 8)
    assert status(bytes(small)) == C.OVERFLOW
    mid = bytearray(raw); mid[8:12] = struct.pack(">i", 20)  # 9 <= bl < 49, enough bytes
    assert status(bytes(mid)) == C.OVERFLOW
    short = bytearray(raw[:40]); short[8:12] = struct.pack(">i", 20)  # 40 bytes of header missing
    assert status(bytes(short)) == C.TRY_GET
    for code in (5, 6, 7):
        bad = bytearray(raw); bad[21:23] = struct.pack(">h", code)
        assert status(bytes(bad)) == C.UNKNOWN_COMPRESSION
    gz = bytearray(raw); gz[21:23] = struct.pack(">h", 1)
    assert status(bytes(gz)) == C.COMPRESSED
    # a wrong CRC is not an error (deflated.rs:164-166); the wire value is kept
    bad_crc = bytearray(raw); bad_crc[17] ^= 0xFF
    st, cols, _, _ = oracle.decode_columns(bytes(bad_crc), offsets_of([bytes(bad_crc)]))
    assert int(cols["status"][0]) == C.OK
    assert int(cols["crc"][0]) != int(cols["crc_computed"][0]) == 1126819645


def test_record_errors():
    raw = bytearray(V["batch_71"]["bytes"])
    def status(data: bytes, count=1):
        buf = bytearray(raw[:61]) + data
        buf[8:12] = struct.pack(">i", 49 + len(data))
        buf[57:61] = struct.pack(">I", count)
        st, cols, _, _ = oracle.decode_columns(bytes(buf), offsets_of([bytes(buf)]))
        return int(cols["status"][0])
    good = bytes([18, 0, 0, 0, 1, 6, 100, 101, 102, 0])
    assert status(good) == C.OK
    assert status(good + b"junk") == C.OK            # trailing bytes are ignored
    assert status(good, 2) == C.TRY_GET              # second record: empty buffer
    assert status(good[:1]) == C.MALFORMED           # get_u8 on empty panics
    assert status(good[:2]) == C.TRY_GET
    assert status(bytes([18, 0, 0, 0, 1, 60, 100])) == C.MALFORMED   # value past the end
    assert status(bytes([18, 0, 0, 0, 3, 1, 0])) == C.MALFORMED      # key length -2
    assert status(bytes([18, 0, 0, 0, 1, 1, 1])) == C.MALFORMED      # header count -1
    assert status(bytes([18, 0, 0, 0, 1, 1, 4])) == C.TRY_GET        # 2 headers, no bytes
    assert status(bytes([0x80] * 5 + [0])) == C.OVERFLOW
    # a length that disagrees with the fields is accepted and kept verbatim
    odd = bytes([40, 0, 0, 0, 1, 6, 100, 101, 102, 0])
    buf = bytearray(raw[:61]) + odd
    st, cols, _, _ = oracle.decode_columns(bytes(buf), offsets_of([bytes(buf)]))
    assert int(cols["status"][0]) == C.OK and int(cols["length"][0]) == 20
    st, out, _ = oracle.encode_columns(cols, 1, bytes(buf), cols["data_off"])
    assert out[61:] == odd


def test_frame_split_and_prefix():
    import ctypes
    raw = bytes(V["batch_71"]["bytes"])
    st, off = oracle.frame_split(raw * 3)
    assert st == C.OK and list(off) == [0, 71, 142, 213]
    assert oracle.frame_split(raw + raw[:30])[0] == C.OVERFLOW
    assert oracle.frame_split(raw + raw[:10])[0] == C.TRY_GET
    assert oracle.frame_split(b"")[0] == C.OK
    buf = ctypes.create_string_buffer(8)
    n = oracle.LIB.no_records_prefix_encode(71, 1, buf)
    assert buf.raw[:n] == bytes([72])
    n = oracle.LIB.no_records_prefix_encode(71, 0, buf)
    assert buf.raw[:n] == bytes([0, 0, 0, 71])
    # the produce v9 frame of the reference carries uvarint(len + 1) = 72 before the batch
    case = [c for c in V["decode_frames"] if c["name"] == "produce_request_v9_000"][0]
    frame = bytes(case["frame"])
    (a, b), = find_batches(frame)
    assert frame[a - 1] == 72 and b - a == 71
    pos = ctypes.c_size_t(a - 1); ln = ctypes.c_uint32(); null = ctypes.c_int()
    assert oracle.LIB.no_records_prefix_decode(frame, len(frame), 1, ctypes.byref(pos),
                                               ctypes.byref(ln), ctypes.byref(null)) == C.OK
    assert (ln.value, null.value, pos.value) == (71, 0, a)


def test_builder_length_quirk():
    # Builder::build sizes each record's prefix as an UNSIGNED varint of the body
    # size, so bodies of 64..127 bytes are under-counted by one byte
    for vlen, short_by in ((10, 0), (70, 1), (200, 0)):
        rec = C.Record(0, 0, 0, 0, None, bytes(vlen))
        cols, payload, src_off = C.columns_from_records([(dict(magic=2), [rec])])
        st, bl, crc = oracle.builder_columns(cols, 1, payload, src_off)
        body = 1 + 1 + 1 + 1 + oracle.varint_size(vlen) + vlen + 1
        cols["length"][0] = body
        st, out, _ = oracle.encode_columns(cols, 1, payload, src_off)
        wire_bl = struct.unpack_from(">i", out, 8)[0]
        assert wire_bl - int(bl[0]) == short_by
        assert struct.unpack_from(">I", out, 17)[0] == int(crc[0])


def test_lean_decode_gives_the_faithful_decode():
    """The decode that fills the columns without the intermediate Vec<Record> (the
    second CPU baseline of bench.py) against the restatement of the reference's:
    golden frames, a seeded workload with headers, and that workload with bytes
    changed at random (batches that fail keep their status and leave no records)."""
    import numpy as np
    from tansu_amd import synth
    from util import find_batches, load_vectors, offsets_of
    blobs = []
    for case in load_vectors()["decode_frames"]:
        frame = bytes(case["frame"])
        batches = [frame[a:b] for a, b in find_batches(frame)]
        blobs.append((np.frombuffer(b"".join(batches), dtype=np.uint8), offsets_of(batches)))
    shape, counts, seed = synth.config("c2", 0.002)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    blobs.append((blob, off))
    rng = np.random.default_rng(9)
    for _ in range(30):
        bad = blob.copy()
        for at in rng.integers(0, len(bad), 40):
            bad[at] = rng.integers(0, 256)
        blobs.append((bad, off))
    for data, offs in blobs:
        for by_value in (True, False):
            st1, a, nr1, nh1 = oracle.decode_columns(data, offs, by_value)
            st2, b, nr2, nh2 = oracle.decode_columns(data, offs, by_value, lean=True)
            assert (st1, nr1, nh1) == (st2, nr2, nh2)
            for name in a:
                assert (a[name] == b[name]).all(), name
