"""The CPU restatement of the two callers of the codec in nisshi-storage: the
produce seam of the SQL engines (pg.rs:797-943, lite.rs:703-815) and combine
(batch.rs:661-709), pinned by the reference's own tests of combine
(batch.rs:1198-1349) and by the golden batches."""
import struct

import numpy as np

import oracle
from tansu_amd import columns as C
from util import build_batch, load_vectors, offsets_of, record_from_json

V = load_vectors()


def decode_one(batch: bytes):
    st, cols, nr, nh = oracle.decode_columns(batch, offsets_of([batch]), True)
    assert st == C.OK and cols["status"][0] == C.OK
    return cols, C.records_of_batch(cols, 0, batch)


# ---------------------------------------------------------------- combine


def into_batches(attributes, producer_id, producer_epoch, base_offset, batches, timestamps=None):
    """batch.rs:1204-1242: one batch per list of values, offset_delta = index,
    last_offset_delta = len - 1, base_sequence running."""
    out = []
    base_sequence = 0
    for i, values in enumerate(batches):
        ts = 0 if timestamps is None else timestamps[i]
        hdr = dict(attributes=attributes, producer_id=producer_id, producer_epoch=producer_epoch,
                   base_offset=base_offset, last_offset_delta=len(values) - 1,
                   base_sequence=base_sequence, base_timestamp=ts, max_timestamp=ts)
        recs = [C.Record(0, 0, 0, od, None, v) for od, v in enumerate(values)]
        out.append(build_batch(hdr, recs))
        base_sequence += len(values)
    return out


REF_BATCHES = [[b"a", b"b", b"c"], [b"f", b"g"], [b"i"], [b"j"], [b"k"], [b"p", b"q", b"r", b"s"]]


def test_combine_empty():
    # batch.rs:1198-1202
    assert oracle.combine([]) == (C.OK, None)


def test_combine_batches():
    # batch.rs:1244-1349
    producer_id, producer_epoch, base_offset, attributes = 54345, 32123, 0, 0
    st, combined = oracle.combine(into_batches(attributes, producer_id, producer_epoch, base_offset,
                                               REF_BATCHES))
    assert st == C.OK
    cols, records = decode_one(combined)
    assert int(cols["producer_id"][0]) == producer_id
    assert int(cols["producer_epoch"][0]) == producer_epoch
    assert int(cols["base_sequence"][0]) == 0
    assert int(cols["base_offset"][0]) == base_offset
    assert int(cols["attributes"][0]) == attributes
    want = [v for values in REF_BATCHES for v in values]
    assert len(records) == len(want) == 12
    for index, (rec, value) in enumerate(zip(records, want)):
        assert rec.key is None and rec.value == value and rec.offset_delta == index
    # what the reference's test does not assert but its code does (batch.rs:697-698, 706)
    assert int(cols["last_offset_delta"][0]) == 11
    assert int(cols["record_count"][0]) == 12
    assert int(cols["crc"][0]) == int(cols["crc_computed"][0])


def test_combine_rebases_timestamps_and_keeps_length():
    # batch.rs:690-691 adds sink.base_timestamp - batch.base_timestamp to every delta;
    # `..record` keeps the stored length although the varints may change size
    ts = [1_000, 1_250, 900, 1_000, 70_000, 5]
    batches = into_batches(0, 1, 2, 40, REF_BATCHES, ts)
    st, combined = oracle.combine(batches)
    assert st == C.OK
    cols, records = decode_one(combined)
    assert int(cols["base_timestamp"][0]) == 1_000 and int(cols["max_timestamp"][0]) == 70_000
    k = 0
    for values, t in zip(REF_BATCHES, ts):
        for _ in values:
            assert records[k].timestamp_delta == 1_000 - t
            assert records[k].length == 7      # as built: one-byte varints, one-byte value
            k += 1
    # a single batch passes through (its records are re-encoded as they are)
    st, one = oracle.combine(batches[:1])
    assert st == C.OK and one == batches[0]


def test_combine_normalises_attributes_and_fails_as_a_whole():
    batches = into_batches(0x0100 | 0x30, 1, 2, 0, REF_BATCHES[:2])
    st, combined = oracle.combine(batches)
    assert st == C.OK
    assert struct.unpack_from(">h", combined, 21)[0] == 0x30      # inflated.rs:123-140
    bad = bytearray(batches[1])
    bad[-1:] = b""                                                # the last record under-runs
    bad[8:12] = struct.pack(">i", struct.unpack_from(">i", bad, 8)[0] - 1)
    st, combined = oracle.combine([batches[0], bytes(bad)])
    assert st != C.OK and combined is None
    compressed = bytearray(batches[0])
    compressed[22] |= 1
    assert oracle.combine([bytes(compressed), batches[1]])[0] == C.COMPRESSED


# ---------------------------------------------------------------- produce rows


def check_rows(batch: bytes, high: int, check_time=True):
    st, attributes, rows, hrows = oracle.produce_rows(batch, high, check_time)
    assert st == C.OK
    cols, records = decode_one(batch)
    assert attributes == int(cols["attributes"][0])
    assert len(rows["offset"]) == len(records)
    span = lambda pos, n: None if n < 0 else batch[int(pos):int(pos) + int(n)]
    h = 0
    for i, rec in enumerate(records):
        assert int(rows["offset"][i]) == high + i                                    # pg.rs:868-869
        assert int(rows["timestamp"][i]) == int(cols["base_timestamp"][0]) + rec.timestamp_delta
        assert span(rows["key_pos"][i], rows["key_len"][i]) == rec.key
        assert span(rows["val_pos"][i], rows["val_len"][i]) == rec.value
        for hd in rec.headers:
            assert int(hrows["offset"][h]) == high + i                               # pg.rs:913-914
            assert span(hrows["key_pos"][h], hrows["key_len"][h]) == hd.key
            assert span(hrows["val_pos"][h], hrows["val_len"][h]) == hd.value
            h += 1
    assert h == len(hrows["offset"])
    return rows, hrows


def test_produce_rows_of_golden_batches():
    check_rows(bytes(V["batch_71"]["bytes"]), 0)
    check_rows(bytes(V["batch_71"]["bytes"]), 32123)
    n = 0
    for case in V["decode_frames"]:
        from util import find_batches
        frame = bytes(case["frame"])
        for a, b in find_batches(frame):
            rows, hrows = check_rows(frame[a:b], 1_000_000)
            n += len(rows["offset"])
    assert n > 0


def test_produce_rows_headers_and_nulls():
    recs = [C.Record(0, 0, 5, 0, b"k0", b"v0", [C.Header(b"a", b"1"), C.Header(b"b", None)]),
            C.Record(0, 0, 6, 1, None, None, []),
            C.Record(0, 0, -3, 2, b"", b"x" * 300, [C.Header(None, b"z")])]
    batch = build_batch(dict(base_timestamp=1_700_000_000_000, max_timestamp=1_700_000_000_006,
                             last_offset_delta=2), recs)
    rows, hrows = check_rows(batch, 77)
    assert list(hrows["offset"]) == [77, 77, 79]
    assert list(rows["timestamp"]) == [1_700_000_000_005, 1_700_000_000_006, 1_699_999_999_997]


def test_produce_rows_negative_timestamp():
    recs = [C.Record(0, 0, 0, 0, None, b"a"), C.Record(0, 0, -11, 1, None, b"b")]
    batch = build_batch(dict(base_timestamp=10, max_timestamp=10, last_offset_delta=1), recs)
    # pg.rs:876 -> to_system_time -> u64::try_from fails (lib.rs:2189-2193)
    assert oracle.produce_rows(batch, 0, True)[0] == C.TRY_FROM_INT
    # lite.rs:791 stores the sum as it is
    st, _, rows, _ = oracle.produce_rows(batch, 0, False)
    assert st == C.OK and list(rows["timestamp"]) == [10, -1]


def test_produce_rows_of_a_batch_that_does_not_inflate():
    raw = bytearray(V["batch_71"]["bytes"])
    assert oracle.produce_rows(bytes(raw[:18]), 0)[0] == C.TRY_GET
    assert oracle.produce_rows(bytes(raw[:30]), 0)[0] == C.OVERFLOW       # deflated.rs:151-153
    raw[22] |= 2
    assert oracle.produce_rows(bytes(raw), 0)[0] == C.COMPRESSED
