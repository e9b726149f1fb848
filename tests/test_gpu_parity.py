"""Parity of the HIP codec (through the C ABI) with the CPU restatement, on the
reference's golden vectors, on seeded synthetic inputs, on malformed inputs, and
through size-independent properties at full size.  Bit-exact: this is integer
and byte work."""
import struct

import numpy as np
import pytest

import oracle
from tansu_amd import columns as C, hip, synth
from util import assert_same, build_batch, find_batches, load_vectors, offsets_of, record_from_json

pytestmark = pytest.mark.gpu

V = load_vectors()


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def gpu_decode(codec, blob, off, by_value=True, rec_cap=None, hdr_cap=None):
    blob = np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
    n_b = len(off) - 1
    d_blob, d_off = codec.to_device(blob, off)
    rec_cap = codec.count_dev(d_blob, len(blob), d_off, n_b) if rec_cap is None else rec_cap
    hdr_cap = len(blob) // 2 + 1 if hdr_cap is None else hdr_cap
    cols = hip.DeviceColumns(n_b, rec_cap, hdr_cap, codec.device)
    totals = codec.decode_dev(d_blob, len(blob), d_off, n_b, cols, by_value)
    return cols, totals, d_blob, d_off


def check_against_oracle(codec, blob, off, by_value=True):
    blob = np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
    n_b = len(off) - 1
    cols, totals, d_blob, d_off = gpu_decode(codec, blob, off, by_value)
    g = cols.to_host()
    st, c, nr, nh = oracle.decode_columns(blob, off, by_value)
    assert st == C.OK
    assert_same(g, c, n_b)
    assert totals.n_failed == int((c["status"] != C.OK).sum())
    return g, c, totals, cols, d_blob, d_off


# ---------------------------------------------------------------- CRC-32C


def test_crc32c_known_answers(codec):
    import torch
    for data in (b"123456789", bytes(V["crc_known_answer"]["bytes"]), b"a", b"ab", b"abc", b"abcd"):
        t = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(codec.device)
        assert codec.crc32c_dev(t, len(data)) == oracle.crc32c(data)
    assert oracle.crc32c(b"123456789") == 0xE3069283


def test_crc32c_lengths_and_alignments(codec):
    import torch
    rng = np.random.default_rng(7)
    buf = rng.integers(0, 256, size=300_000, dtype=np.uint8)
    t = torch.from_numpy(buf).to(codec.device)
    lengths = [1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 1000, 1023, 1024, 1025, 2047, 2048, 4096, 5000,
               65535, 65536, 65537, 131072, 200_001]
    for n in lengths:
        for start in (0, 1, 7, 15, 16, 21, 1003):
            if start + n > len(buf):
                continue
            assert codec.crc32c_dev(t, n, offset=start) == oracle.crc32c(buf[start:start + n].tobytes()), (n, start)


def test_crc32c_zero_runs(codec):
    # long zero runs exercise the fold with an all-zero accumulator and the padding inverse
    import torch
    for n in (1, 1024, 70_000):
        z = torch.zeros(n, dtype=torch.uint8, device=codec.device)
        assert codec.crc32c_dev(z, n) == oracle.crc32c(bytes(n))


# ---------------------------------------------------------------- golden vectors


@pytest.mark.parametrize("case", V["decode_frames"], ids=lambda c: c["name"])
def test_golden_decode_frames(codec, case):
    frame = bytes(case["frame"])
    spans = find_batches(frame)
    blob = b"".join(frame[a:b] for a, b in spans)
    off = hip.frame_split(blob)
    g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, blob, off)
    for i, want in enumerate(case["batches"]):
        assert int(g["status"][i]) == C.OK
        for k, w in want.items():
            if k != "records":
                assert int(g[k][i]) == w, (i, k)
        assert int(g["crc_computed"][i]) == want["crc"]
        assert C.records_of_batch(g, i, blob) == [record_from_json(r) for r in want["records"]]
    # bit-exact re-encode on the GPU
    import torch
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(len(spans) + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, len(spans), d_blob, cols.t["data_off"], out, len(blob), out_off)
    assert t.out_bytes == len(blob)
    assert out[:len(blob)].cpu().numpy().tobytes() == blob
    assert list(out_off.cpu().numpy().view(np.uint64)) == list(off)


@pytest.mark.parametrize("case", V["encode_frames"] + V["bench_frames"], ids=lambda c: c["name"])
def test_golden_wire_frames_round_trip(codec, case):
    frame = bytes(case["frame"])
    spans = find_batches(frame)
    blob = b"".join(frame[a:b] for a, b in spans)
    off = hip.frame_split(blob)
    g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, blob, off)
    assert (g["crc"] == g["crc_computed"]).all() and totals.n_failed == 0
    cols_h, t2 = codec.decode(blob, off)               # host-buffer entry point
    out, out_off, t3 = codec.encode(cols_h, blob, cols_h["data_off"])
    assert out == blob


def test_golden_small_vectors(codec):
    raw = bytes(V["batch_71"]["bytes"])
    g, c, *_ = check_against_oracle(codec, raw, offsets_of([raw]))
    assert C.records_of_batch(g, 0, raw) == [record_from_json(r) for r in V["batch_71"]["batch"]["records"]]
    v = V["control_batch"]
    hdr = v["batch"]
    data = bytes(v["record_data"])
    ctrl = struct.pack(">qiibIhiqqqhiI", hdr["base_offset"], hdr["batch_length"],
                       hdr["partition_leader_epoch"], hdr["magic"], hdr["crc"], hdr["attributes"],
                       hdr["last_offset_delta"], hdr["base_timestamp"], hdr["max_timestamp"],
                       hdr["producer_id"], hdr["producer_epoch"], hdr["base_sequence"],
                       hdr["record_count"]) + data
    g, c, *_ = check_against_oracle(codec, ctrl, offsets_of([ctrl]))
    (rec,) = C.records_of_batch(g, 0, ctrl)
    assert rec.key == bytes(v["key"]) and rec.value == bytes(v["value"])
    assert int(g["crc_computed"][0]) == hdr["crc"]


def test_by_reference_keeps_attributes(codec):
    raw = bytearray(V["batch_71"]["bytes"])
    raw[21:23] = struct.pack(">h", 0x0100 | 0x78)       # unknown bit 8 + flag bits, no compression
    raw = bytes(raw)
    g, c, *_ = check_against_oracle(codec, raw, offsets_of([raw]), by_value=True)
    assert int(g["attributes"][0]) == 0x78
    g, c, *_ = check_against_oracle(codec, raw, offsets_of([raw]), by_value=False)
    assert int(g["attributes"][0]) == 0x0178


# ---------------------------------------------------------------- synthetic inputs vs oracle


def crc_patch(codec, blob, off):
    cols, totals, d_blob, d_off = gpu_decode(codec, blob, off, rec_cap=1, hdr_cap=1)
    crc = cols.t["crc_computed"].cpu().numpy().view(np.uint32)[:len(off) - 1]
    synth.patch_crc(blob, off, crc)
    return crc


@pytest.mark.parametrize("name,scale", [("c1", 0.02), ("c2", 0.004), ("c3", 0.02), ("c5", 0.02)])
def test_synthetic_vs_oracle(codec, name, scale):
    shape, counts, seed = synth.config(name, scale)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    crc_patch(codec, blob, off)
    g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, blob, off)
    assert totals.n_failed == 0 and totals.n_exact == 0
    assert totals.n_records == nr and nh <= totals.n_headers <= nh + 4096 * len(counts)
    assert int(g["hdr_count"].sum()) == nh
    assert (g["crc"] == g["crc_computed"]).all()
    # re-encode on the GPU equals the oracle's re-encode and the (canonical) input
    import torch
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(len(counts) + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, len(counts), d_blob, cols.t["data_off"], out, len(blob), out_off)
    got = out[:t.out_bytes].cpu().numpy()
    st, want, _ = oracle.encode_columns(c, len(counts), blob, c["data_off"])
    assert got.tobytes() == want == blob.tobytes()


def test_records_of_changing_sizes_vs_oracle(codec):
    """C2 with value and header-value lengths drawn per record: no two neighbours
    have the same size, so record starts can only be found by following the length
    varints.  Columns against the oracle, then the re-encode against the input."""
    import torch
    blob, off, nr, nh = synth.make_var_batches(24, 1000, synth.SEED_BASE + 6)
    crc_patch(codec, blob, off)
    g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, blob, off)
    assert (totals.n_failed, totals.n_exact, totals.n_records) == (0, 0, nr)
    assert int(g["hdr_count"].sum()) == nh and (g["crc"] == g["crc_computed"]).all()
    assert len(np.unique(c["val_len"])) == 65 and len(np.unique(c["hval_len"])) == 33
    n_b = len(off) - 1
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, n_b, d_blob, cols.t["data_off"], out, len(blob), out_off)
    assert out[:t.out_bytes].cpu().numpy().tobytes() == blob.tobytes()


def test_changing_sizes_with_payloads_that_look_like_record_heads():
    """The chained walk of large batches looks for places where a record may start (a length of two bytes
    that is not negative, then a zero byte) and lets hops from the round's first record decide which of them
    are: payloads made of exactly such bytes, records of a byte's length among them, records with nonzero
    attributes (which no candidate is found for), and a batch of nothing but short records."""
    import os
    os.environ["NISSHI_WIDE_RING_FROM"] = "0"      # (the kernel of large batches, whatever the batches' size)
    try:
        codec = hip.Codec(0)
    finally:
        del os.environ["NISSHI_WIDE_RING_FROM"]
    rng = np.random.default_rng(41)
    head = bytes([0x90, 0x01, 0x00])               # as a record's first bytes: length 72, attributes 0
    batches = []
    for kind in range(5):
        records = []
        for i in range(1100):
            n = int(rng.integers(60, 200))
            value = (head * 80)[int(rng.integers(0, 3)):][:n]
            key = None if i % 7 else (head * 4)[:int(rng.integers(0, 12))]
            headers = [C.Header(b"h", (head * 9)[:int(rng.integers(0, 25))])] if i % 3 == 0 else []
            attributes = 0
            if kind == 1 and i % 50 == 49:
                value = value[:int(rng.integers(1, 30))]          # a record under 64 bytes: a length of one byte
            if kind == 2 and i % 40 == 11:
                attributes = 0x20
            if kind == 3:
                value = value[:20]                                # all short
            if kind == 4 and i % 200 == 150:
                value = bytes(rng.integers(0, 256, 9000, dtype=np.uint8))   # a length of three bytes
            records.append(C.Record(0, attributes, i * 3, i, key, value, headers))
        batches.append(build_batch(dict(base_offset=kind * 5000, last_offset_delta=len(records) - 1), records))
    blob = b"".join(batches)
    off = offsets_of(batches)
    g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, blob, off)
    assert (totals.n_failed, totals.n_records) == (0, 5 * 1100)
    codec.close()


def test_changing_sizes_with_a_wrong_length_and_a_large_record(codec):
    """The chained walk's ways out: a record whose length disagrees with its fields
    deep inside a batch (the sequential walk takes over and accepts it, as the
    reference does), a record larger than the LDS ring between small ones (the walk
    from global memory takes over), and a batch whose count runs past its data."""
    rng = np.random.default_rng(17)

    def recs(n, big_at=None):
        out = []
        for i in range(n):
            v = bytes(rng.integers(0, 256, 20000 if i == big_at else int(rng.integers(0, 120)), dtype=np.uint8))
            hdrs = [C.Header(b"k%d" % j, bytes(rng.integers(0, 256, int(rng.integers(0, 12)), dtype=np.uint8)))
                    for j in range(int(rng.integers(0, 4)))]
            out.append(C.Record(0, 0, i, i, None if i % 7 == 0 else b"key%d" % i, v, hdrs))
        return out

    plain = build_batch(dict(base_offset=0, last_offset_delta=299), recs(300))
    big = build_batch(dict(base_offset=300, last_offset_delta=399), recs(400, big_at=250))
    wrong = bytearray(build_batch(dict(base_offset=700, last_offset_delta=199), recs(200)))
    # the length varint of a record far inside: find record 150's start through the oracle
    st, c, _, _ = oracle.decode_columns(bytes(wrong), offsets_of([bytes(wrong)]))
    at = 61 + int(c["key_off"][150]) - 1          # its key length varint (one byte: short keys)
    while wrong[at] != (2 * len(b"key150")):      # step back over ts/od varints to be safe
        at -= 1
    rec_start = at - 1 - len(synth.varint(150)) * 2 - len(synth.varint(int(c["length"][150])))
    assert len(synth.varint(int(c["length"][150]))) == 2
    wrong[rec_start] ^= 0x02                      # length changes by one: it now disagrees with the fields
    short = bytearray(build_batch(dict(base_offset=900, last_offset_delta=99), recs(100)))
    short[57:61] = struct.pack(">I", 101)         # one record more than the data holds
    batches = [plain, big, bytes(wrong), bytes(short), plain]
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8)
    g, c, totals, *_ = check_against_oracle(codec, blob, offsets_of(batches))
    assert list(c["status"]) == [C.OK, C.OK, C.OK, C.TRY_GET, C.OK]
    assert totals.n_exact >= 2 and totals.n_failed == 1


def test_mixed_partitions_round_trip(codec):
    import torch
    for blob, off, nr, nh in synth.mixed_partitions(10, 3):
        crc_patch(codec, blob, off)
        g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, blob, off)
        assert totals.n_failed == 0
        out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
        out_off = torch.zeros(len(off), dtype=torch.int64, device=codec.device)
        t = codec.encode_dev(cols, len(off) - 1, d_blob, cols.t["data_off"], out, len(blob), out_off)
        assert out[:t.out_bytes].cpu().numpy().tobytes() == blob.tobytes()


# ---------------------------------------------------------------- odd and malformed inputs


def _batch(data: bytes, count: int, attributes=0, batch_length=None) -> bytes:
    bl = 49 + len(data) if batch_length is None else batch_length
    return struct.pack(">qiibIhiqqqhiI", 7, bl, -1, 2, 0, attributes, 0, 1, 2, -1, -1, -1, count) + data


GOOD = bytes([18, 0, 0, 0, 1, 6, 100, 101, 102, 0])


def odd_batches():
    rec_hdr = bytes([26, 0, 2, 4, 2, 107, 2, 118, 2, 2, 97, 4, 98, 99])  # 1 header (a -> bc)
    out = [
        _batch(GOOD, 1), _batch(GOOD * 3, 3), _batch(GOOD + b"junk", 1), _batch(GOOD, 2),
        _batch(b"", 0), _batch(b"", 1), _batch(GOOD[:1], 1), _batch(GOOD[:2], 1),
        _batch(bytes([18, 0, 0, 0, 1, 60, 100]), 1),        # value runs past the end
        _batch(bytes([18, 0, 0, 0, 3, 1, 0]), 1),           # key length -2
        _batch(bytes([18, 0, 0, 0, 1, 1, 1]), 1),           # header count -1
        _batch(bytes([18, 0, 0, 0, 1, 1, 4]), 1),           # 2 headers, no bytes
        _batch(bytes([0x80] * 5 + [0]), 1),                 # 6-byte varint
        _batch(bytes([0x80] * 5), 1),                       # truncated varint
        _batch(bytes([40, 0, 0, 0, 1, 6, 100, 101, 102, 0]), 1),        # length disagrees (too long)
        _batch(bytes([2, 0, 0, 0, 1, 6, 100, 101, 102, 0]) + GOOD, 2),  # length disagrees (too short)
        _batch(bytes([0x92, 0x80, 0x00, 0, 0x80, 0x00, 0, 1, 6, 100, 101, 102, 0]), 1),  # padded varints
        _batch(bytes([18, 0, 0, 0, 1, 6, 100, 101, 102, 0x80, 0x00]), 1),  # padded header count
        _batch(rec_hdr, 1), _batch(rec_hdr * 70, 70),
        _batch(bytes([1, 0]), 1),                           # negative length -1, then nothing
        _batch(GOOD, 0xFFFFFFFF), _batch(GOOD * 2, 3),
        _batch(GOOD, 1, attributes=5), _batch(GOOD, 1, attributes=1), _batch(GOOD, 1, attributes=0x48),
        _batch(GOOD, 1, batch_length=-5), _batch(GOOD, 1, batch_length=8),
        _batch(GOOD, 1, batch_length=20), _batch(GOOD, 1, batch_length=1000),
        _batch(GOOD, 1)[:20], _batch(GOOD, 1)[:40], _batch(GOOD, 1)[:7], b"",
        # timestamp varlong of 10 bytes, 11 bytes (overflow), and i64::MIN
        _batch(bytes([36, 0]) + bytes([0xFF] * 9 + [0x01]) + bytes([0, 1, 6, 100, 101, 102, 0]), 1),
        _batch(bytes([38, 0]) + bytes([0x80] * 10 + [0x00]) + bytes([0, 1, 6, 100, 101, 102, 0]), 1),
        # offset_delta with 5 bytes whose upper bits are dropped
        _batch(bytes([26, 0, 0, 0xFF, 0xFF, 0xFF, 0xFF, 0x7F, 1, 6, 100, 101, 102, 0]), 1),
    ]
    return out


def test_odd_and_malformed_batches(codec):
    batches = odd_batches()
    blob = b"".join(batches)
    off = offsets_of(batches)
    for by_value in (True, False):
        g, c, totals, *_ = check_against_oracle(codec, blob, off, by_value)
        assert totals.n_exact >= 2
    st = dict(zip(range(len(batches)), c["status"]))
    assert st[0] == C.OK and st[3] == C.TRY_GET and st[6] == C.MALFORMED and st[12] == C.OVERFLOW
    assert st[14] == C.OK and st[15] == C.OK            # accepted by the sequential walk


def test_small_batches_a_lane_each_and_a_workgroup_each(codec):
    """A blob whose batches average under 1 KiB is decoded a lane a batch; the same
    batches beside one large batch (which lifts the average) a workgroup a batch.
    Both must give the oracle's columns, malformed batches included."""
    from util import load_vectors, find_batches
    V = load_vectors()
    small = odd_batches()
    for case in V["decode_frames"]:
        frame = bytes(case["frame"])
        small += [frame[a:b] for a, b in find_batches(frame)]
    shape, counts, seed = synth.config("c2", 1 / 8192)
    big, big_off, _, _ = synth.make_batches(shape, counts, seed)
    big = bytes(big)
    assert len(b"".join(small)) / len(small) < 1024 < (len(b"".join(small)) + len(big)) / (len(small) + 1)
    ga, ca, ta, *_ = check_against_oracle(codec, b"".join(small), offsets_of(small))
    both = small + [big]
    gb, cb, tb, *_ = check_against_oracle(codec, b"".join(both), offsets_of(both))
    n = len(small)
    for name in ("status", "crc_computed", "record_count", "hdr_count", "attributes", "base_offset"):
        assert (ga[name][:n] == gb[name][:n]).all(), name
    # the lane-a-batch path counts as sequential only what had to be (a count the data cannot hold)
    assert ta.n_exact <= tb.n_exact


def test_reencode_of_non_canonical_input(codec):
    # padded varints decode fine and re-encode shorter; the stored length is emitted verbatim
    import torch
    batches = [odd_batches()[i] for i in (14, 16, 17, 19)]
    blob = b"".join(batches)
    off = offsets_of(batches)
    g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, blob, off)
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(len(off), dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, len(batches), d_blob, cols.t["data_off"], out, len(blob) + 64, out_off)
    st, want, want_off = oracle.encode_columns(c, len(batches), blob, c["data_off"])
    assert out[:t.out_bytes].cpu().numpy().tobytes() == want
    assert want != blob and len(want) < len(blob)


def test_mutation_fuzz(codec):
    """Bit flips, truncations and count changes of valid batches: statuses and
    every accepted record must match the sequential CPU walk; nothing may fault."""
    rng = np.random.default_rng(20240607)
    shape = synth.Shape(key_len=5, val_len=20, n_headers=1, hkey_len=2, hval_len=3)
    base, boff, _, _ = synth.make_batches(shape, np.full(1, 12), 99)
    base = base.tobytes()
    batches = []
    for i in range(1500):
        b = bytearray(base)
        kind = i % 5
        if kind == 0:
            for _ in range(rng.integers(1, 4)):
                p = rng.integers(61, len(b))
                b[p] ^= 1 << rng.integers(0, 8)
        elif kind == 1:
            p = rng.integers(61, len(b))
            b[p] = rng.integers(0, 256)
        elif kind == 2:
            cut = rng.integers(0, len(b))
            b = b[:cut]
        elif kind == 3:
            b[57:61] = struct.pack(">I", int(rng.integers(0, 40)))
        else:
            p = rng.integers(8, 61)
            b[p] = rng.integers(0, 256)
        batches.append(bytes(b))
    blob = b"".join(batches)
    off = offsets_of(batches)
    g, c, totals, *_ = check_against_oracle(codec, blob, off)
    assert 0 < totals.n_failed < len(batches)


def test_capacity_is_reported(codec):
    shape, counts, seed = synth.config("c2", 0.001)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    cols, totals, *_ = gpu_decode(codec, blob, off, rec_cap=nr - 1, hdr_cap=nh)
    st = cols.to_host(0, 0)["status"]
    assert st[-1] == C.CAPACITY and (st[:-1] == C.OK).all() and totals.n_failed == 1
    cols, totals, *_ = gpu_decode(codec, blob, off, rec_cap=nr, hdr_cap=nh - 1)
    st = cols.to_host(0, 0)["status"]
    assert (st == C.CAPACITY).sum() == 1 and totals.n_failed == 1 and totals.n_headers >= nh


def test_empty_call(codec):
    cols, totals, *_ = gpu_decode(codec, b"", np.zeros(1, dtype=np.uint64), rec_cap=1, hdr_cap=1)
    assert (totals.n_records, totals.n_headers, totals.n_failed) == (0, 0, 0)


# ---------------------------------------------------------------- full size: properties


@pytest.mark.parametrize("name", ["c1", "c2", "c3"])
def test_full_size_properties(codec, name):
    """At the full benchmark sizes the oracle is too slow to run in a test, so check
    what the domain guarantees: every CRC matches after patching with the GPU's own
    CRC (itself checked against the oracle on a sample of batches), record and
    header totals, arithmetic structure of the decoded columns, and the
    decode -> re-encode round trip reproducing the input byte for byte."""
    import torch
    shape, counts, seed = synth.config(name, 1.0)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    crc = crc_patch(codec, blob, off)
    for b in (0, len(counts) // 2, len(counts) - 1):
        assert int(crc[b]) == oracle.crc32c(blob[int(off[b]) + 21:int(off[b + 1])].tobytes())
    n_b = len(counts)
    cols, totals, d_blob, d_off = gpu_decode(codec, blob, off, rec_cap=nr, hdr_cap=max(nh, 1))
    assert (totals.n_records, totals.n_headers, totals.n_failed, totals.n_exact) == (nr, nh, 0, 0)
    assert int(cols.t["hdr_count"].sum()) == nh
    t = cols.t
    assert bool((t["status"] == 0).all()) and bool((t["crc"] == t["crc_computed"]).all())
    r = int(counts[0])
    f = cols.field
    od = f("offset_delta")[:nr].reshape(n_b, r)
    assert bool((od == torch.arange(r, device=od.device, dtype=od.dtype)).all())
    assert bool((f("timestamp_delta")[:nr].view(n_b, r) == torch.arange(r, device=od.device)).all())
    assert bool((f("val_len")[:nr] == shape.val_len).all()) and bool((f("key_len")[:nr] == shape.key_len).all())
    if nh:
        # uniform batches use exactly their header count; every record owns its own run of slots
        assert bool((f("hkey_len")[:nh] == shape.hkey_len).all()) and bool((f("hval_len")[:nh] == shape.hval_len).all())
        base = f("rhdr_base")[:nr].to(torch.int64).sort().values
        assert bool((base == torch.arange(nr, device=od.device) * shape.n_headers).all())
    # one batch in full against the oracle
    b = n_b // 3
    sub = blob[int(off[b]):int(off[b + 1])]
    st, c, _, _ = oracle.decode_columns(sub, np.array([0, len(sub)], dtype=np.uint64))
    g = cols.to_host()
    g0 = int(g["rec_base"][b])
    for name_ in ("length", "key_off", "val_off", "timestamp_delta"):
        assert (g[name_][g0:g0 + r] == c[name_]).all()
    del g
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
    te = codec.encode_dev(cols, n_b, d_blob, t["data_off"], out, len(blob), out_off)
    assert te.out_bytes == len(blob)
    assert bool((out[:len(blob)] == d_blob[:len(blob)]).all())


@pytest.mark.parametrize("name", ["c5", "c4"])
def test_full_size_ragged_and_mixed(codec, name):
    """The ragged workload at its full size (batches of up to 10 000 records: the
    walks of large batches at scale) and a GPU's share of the mixed one: totals,
    CRCs, per-record arithmetic (offset and timestamp deltas count up from 0 in
    every batch), the largest batch in full against the oracle, and the re-encode
    reproducing the input byte for byte."""
    import torch
    from tansu_amd import shard
    if name == "c5":
        shape, counts, seed = synth.config("c5", 1.0)
        blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    else:
        _, blob, off, nr, nh, _ = shard.make_shard("c4", 1.0, 0, 1)
    n_b = len(off) - 1
    crc_patch(codec, blob, off)
    cols, totals, d_blob, d_off = gpu_decode(codec, blob, off, rec_cap=nr, hdr_cap=max(nh, 1) + 4096 * 16)
    assert (totals.n_records, totals.n_failed, totals.n_exact) == (nr, 0, 0)
    t = cols.t
    assert int(t["hdr_count"][:n_b].sum()) == nh
    assert bool((t["status"][:n_b] == 0).all()) and bool((t["crc"][:n_b] == t["crc_computed"][:n_b]).all())
    assert int(t["record_count"][:n_b].sum()) == nr
    # record r of a batch has offset_delta = timestamp_delta = its index in the batch
    base = t["rec_base"][:n_b].to(torch.int64)
    count = t["record_count"][:n_b].to(torch.int64)
    assert bool((base == torch.cumsum(count, 0) - count).all())
    index = torch.arange(nr, device=base.device) - torch.repeat_interleave(base, count)
    f = cols.field
    assert bool((f("offset_delta")[:nr] == index).all())
    assert bool((f("timestamp_delta")[:nr] == index).all())
    b = int(torch.argmax(count))
    sub = blob[int(off[b]):int(off[b + 1])]
    st, c, _, _ = oracle.decode_columns(sub, np.array([0, len(sub)], dtype=np.uint64))
    assert st == 0
    g0, r = int(base[b]), int(count[b])
    for name_ in ("length", "key_off", "key_len", "val_off", "val_len", "timestamp_delta", "rhdr_count"):
        assert (f(name_)[g0:g0 + r].cpu().numpy() == c[name_][:r]).all(), name_
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
    te = codec.encode_dev(cols, n_b, d_blob, t["data_off"], out, len(blob), out_off)
    assert te.out_bytes == len(blob)
    assert bool((out[:len(blob)] == d_blob[:len(blob)]).all())


def test_host_api_with_pinned_buffers(codec):
    """The host-buffer entry point reading a page-locked blob and writing page-locked
    columns gives what it gives with ordinary memory: the oracle's columns."""
    shape, counts, seed = synth.config("c2", 8 / 8192)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    crc_patch(codec, blob, off)
    n_b = len(counts)
    p_blob = codec.pinned((len(blob),), np.uint8)
    p_blob[:] = np.frombuffer(bytes(blob), dtype=np.uint8)
    p_off = np.ascontiguousarray(off, dtype=np.uint64)
    arrays = codec.pinned_columns(n_b, nr, nh + 4096 * n_b)
    totals = codec.decode_into(p_blob, p_off, arrays, nr, nh + 4096 * n_b)
    assert (totals.n_records, totals.n_failed) == (nr, 0)
    st, c, _, _ = oracle.decode_columns(blob, off)
    assert st == 0
    from tansu_amd import columns as C
    assert_same(C.unpack(arrays), c, n_b)


def test_encode_of_many_small_batches(codec):
    """From 64 batches with under 4 KiB of output capacity each on, batches of up to
    512 bytes of record data are encoded a
    lane each: the same bytes as the oracle's, keys, nulls and headers included, in
    one call with batches that take the wave-a-batch path."""
    import torch
    rng = np.random.default_rng(3)
    batches = []
    for b in range(150):
        recs = []
        for i in range(int(rng.integers(0, 5))):
            hdrs = [C.Header(b"h%d" % j, None if j % 2 else b"x" * j) for j in range(int(rng.integers(0, 3)))]
            recs.append(C.Record(0, 0, 7 * i - 3, i, None if i % 3 == 0 else b"k" * int(rng.integers(0, 20)),
                                 None if i % 4 == 1 else bytes(rng.integers(0, 256, int(rng.integers(0, 90)), dtype=np.uint8)),
                                 hdrs))
        batches.append(build_batch(dict(base_offset=b, last_offset_delta=max(len(recs) - 1, 0)), recs))
    shape, counts, seed = synth.config("c1", 0.001)               # three batches of 11 KB among them
    big, big_off, _, _ = synth.make_batches(shape, counts, seed)
    big = bytes(big)
    batches[50:50] = [big[int(big_off[i]):int(big_off[i + 1])] for i in range(3)]
    blob, off = b"".join(batches), offsets_of(batches)
    crc_blob = np.frombuffer(blob, dtype=np.uint8).copy()
    crc_patch(codec, crc_blob, off)
    g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, crc_blob, off)
    n_b = len(batches)
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, n_b, d_blob, cols.t["data_off"], out, len(blob), out_off)
    st, want, want_off = oracle.encode_columns(c, n_b, crc_blob, c["data_off"])
    assert st == C.OK
    assert out[:t.out_bytes].cpu().numpy().tobytes() == want == crc_blob.tobytes()
    assert (out_off.cpu().numpy().astype(np.uint64) == want_off).all()


def test_scans_over_more_batches_than_one_workgroup_takes(codec):
    """From 32769 batches on the scans over batches (record slots and CRC work items
    in decode, output offsets and skipped batches in encode) run a workgroup a tile
    of 8192 batches instead of one workgroup for all: same columns and same bytes as
    the oracle's, malformed batches among them, the last tile a partial one."""
    import torch
    small = odd_batches()
    for case in V["decode_frames"]:
        frame = bytes(case["frame"])
        small += [frame[a:b] for a, b in find_batches(frame)]
    n_b = 5 * 8192 + 77
    batches = [small[(i * 7 + i // len(small)) % len(small)] for i in range(n_b)]
    # two batches of 1 MiB values (CRC by chunks on both sides), one of them in the last tile
    shape, counts, seed = synth.config("c3", 2 / 256)
    big, big_off, _, _ = synth.make_batches(shape, counts, seed)
    big = bytes(big)
    for at, i in ((20000, 0), (n_b - 3, 1)):
        one = bytearray(big[int(big_off[i]):int(big_off[i + 1])])
        one[17:21] = struct.pack(">I", oracle.crc32c(bytes(one[21:])))
        batches[at] = bytes(one)
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8)
    off = offsets_of(batches)
    g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, blob, off)
    assert totals.n_failed > 0
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, n_b, d_blob, cols.t["data_off"], out, len(blob) + 64, out_off)
    st, want, want_off = oracle.encode_columns(c, n_b, blob, c["data_off"])
    assert st == C.OK
    assert t.n_failed == totals.n_failed
    assert (out_off.cpu().numpy().astype(np.uint64) == want_off).all()
    assert out[:t.out_bytes].cpu().numpy().tobytes() == want


def test_encode_long_spans_across_crc_chunks(codec):
    """Keys and values from just under 16 KiB to over 1 MiB at every alignment: the
    encoder cuts them at the batch's 64 KiB CRC chunks and takes the CRC of the whole
    chunks while it copies them, the rest from a pass over the output.  Bytes and CRC
    must be the oracle's."""
    import torch
    rng = np.random.default_rng(11)
    sizes = [16383, 16384, 16385, 65535, 65536, 65537, 131071, 131072, 140000, 200001, (1 << 20) + 17]
    batches = []
    for i, n in enumerate(sizes):
        recs = [C.Record(0, 0, 0, 0, b"k" * (i * 3 % 17), bytes(rng.integers(0, 256, n, dtype=np.uint8)), [])]
        if i % 3 == 0:   # a long key as well, a short record between two long ones, headers after
            recs.append(C.Record(0, 0, 1, 1, None, b"short", [C.Header(b"h", b"v" * 700)]))
            recs.append(C.Record(0, 0, 2, 2, bytes(rng.integers(0, 256, 70000 + i, dtype=np.uint8)),
                                 bytes(rng.integers(0, 256, 3 * 65536 + i, dtype=np.uint8)), []))
        batches.append(build_batch(dict(base_offset=i, last_offset_delta=len(recs) - 1), recs))
        batches.append(build_batch(dict(base_offset=1000 + i), [C.Record(0, 0, 0, 0, None, b"x" * (i + 1), [])]))
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8)
    off = offsets_of(batches)
    n_b = len(batches)
    g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, blob, off)
    assert totals.n_failed == 0 and (g["crc"][:n_b] == g["crc_computed"][:n_b]).all()
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, n_b, d_blob, cols.t["data_off"], out, len(blob), out_off)
    assert out[:t.out_bytes].cpu().numpy().tobytes() == blob.tobytes()


@pytest.mark.parametrize("seed", [5, 6, 7, 8])
def test_records_of_many_shapes_in_one_batch(codec, seed):
    """Records that differ from one to the next (null and long keys, values from
    empty to 20 KB, up to 20 headers with null values, negative and five-byte
    deltas, attributes): the walk learns shapes, gives up on them, leaves the LDS
    ring for global memory and comes back, all within a batch."""
    import torch
    rng = np.random.default_rng(seed)

    def blob_of(n):
        return bytes(rng.integers(0, 256, n, dtype=np.uint8))

    batches = []
    for b in range(120):
        recs = []
        style = b % 4                # 0: every record its own shape; 1: runs of equal shapes; 2: mostly equal; 3: mixed sizes
        run_shape = None
        for i in range(int(rng.integers(1, 260))):
            fresh = style == 0 or run_shape is None or (style == 1 and rng.random() < 0.1) or \
                (style == 2 and rng.random() < 0.02) or (style == 3 and rng.random() < 0.3)
            if fresh:
                big = style == 3 and rng.random() < 0.15
                run_shape = (None if rng.random() < 0.3 else int(rng.integers(0, 40)),
                             None if rng.random() < 0.1 else int(rng.integers(7000, 20000) if big else rng.integers(0, 300)),
                             int(rng.integers(0, 21)) if rng.random() < 0.1 else int(rng.integers(0, 3)))
            kl, vl, nh = run_shape
            hdrs = [C.Header(blob_of(int(rng.integers(0, 9))), None if (i + j) % 5 == 0 else blob_of(int(rng.integers(0, 30))))
                    for j in range(nh)]
            ts = int(rng.integers(-5, 5)) if rng.random() < 0.9 else int(rng.integers(-2**40, 2**40))
            od = i if rng.random() < 0.95 else int(rng.integers(-2**31, 2**31))
            recs.append(C.Record(0, int(rng.integers(0, 2)) * 7, ts, od, None if kl is None else blob_of(kl),
                                 None if vl is None else blob_of(vl), hdrs))
        batches.append(build_batch(dict(base_offset=b * 1000, last_offset_delta=len(recs) - 1, attributes=16 * (b % 2)), recs))
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8)
    off = offsets_of(batches)
    n_b = len(batches)
    g, c, totals, cols, d_blob, d_off = check_against_oracle(codec, blob, off)
    assert totals.n_failed == 0 and (g["crc"][:n_b] == g["crc_computed"][:n_b]).all()
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, n_b, d_blob, cols.t["data_off"], out, len(blob), out_off)
    assert out[:t.out_bytes].cpu().numpy().tobytes() == blob.tobytes()
    # where headers go does not depend on the order the batches decode in: two decodes
    # into zeroed columns give the same columns byte for byte, and every header slot set
    # aside is a batch's expected run or the run of a batch that needed more
    twice = []
    for _ in range(2):
        fresh = hip.DeviceColumns(n_b, cols.rec_cap, cols.hdr_cap, codec.device)
        for t_ in fresh.t.values():
            t_.zero_()
        tot = codec.decode_dev(d_blob, len(blob), d_off, n_b, fresh)
        twice.append((fresh.to_host(), tot))
    (g1, t1), (g2, t2) = twice
    assert (t1.n_records, t1.n_headers) == (t2.n_records, t2.n_headers) == (totals.n_records, totals.n_headers)
    for name in g1:
        assert g1[name].tobytes() == g2[name].tobytes(), name
    assert int(g["hdr_count"][:n_b].sum()) <= totals.n_headers <= len(blob) + 1
