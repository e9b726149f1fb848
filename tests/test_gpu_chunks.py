"""Large batches of small records are decoded chunk by chunk, a wave a 64 KiB
chunk, on guesses that are checked afterwards (where a chunk's first record
starts, which record it is, how many headers a record has); a batch whose chunks
do not join up is walked by one wave.  Either way the columns are the sequential
walk's (deflated.rs:431-442 with record.rs:254-277): against the oracle, on batches
built to take every way through."""
import os
import struct

import numpy as np
import pytest

import oracle
from tansu_amd import columns as C, hip, synth
from test_gpu_parity import check_against_oracle, crc_patch
from util import build_batch, offsets_of

pytestmark = pytest.mark.gpu

CHUNK = 65536


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def records(n, rng, value=lambda i: 100, headers=lambda i: 0, key=lambda i: b"key%05d" % i, od=lambda i: i,
            ts=lambda i: i):
    out = []
    for i in range(n):
        v = bytes(rng.integers(0, 256, value(i), dtype=np.uint8))
        h = [C.Header(b"h%d" % j, b"v%d" % (i % 10)) for j in range(headers(i))]
        out.append(C.Record(0, 0, ts(i), od(i), key(i), v, h))
    return out


def decode(codec, batches):
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8).copy()
    off = offsets_of(batches)
    crc_patch(codec, blob, off)
    return check_against_oracle(codec, blob, off)


def test_chunks_join(codec):
    """Records of one kind, 10 000 of them (the offset delta and the timestamp
    delta grow a byte at 64 and at 8192), and a batch whose records change size
    and kind in the middle of a chunk, header count unchanged: every chunk joins
    the next, nothing is walked twice."""
    rng = np.random.default_rng(1)
    same = build_batch(dict(base_offset=0, last_offset_delta=9999), records(10000, rng, headers=lambda i: 2))
    change = build_batch(dict(base_offset=10000, last_offset_delta=5999),
                         records(6000, rng, value=lambda i: 100 if i < 2500 else 31 + i % 7,
                                 key=lambda i: None if i > 4000 else b"k%d" % i))
    assert len(same) > 8 * CHUNK and len(change) > 4 * CHUNK
    g, c, totals, *_ = decode(codec, [same, change, same])
    assert (totals.n_failed, totals.n_exact, totals.n_records) == (0, 0, 26000)
    assert list(g["hdr_count"][:3]) == [20000, 0, 20000]
    assert (g["crc"][:3] == g["crc_computed"][:3]).all()


def test_chunks_that_do_not_join(codec):
    """What the guesses cannot know: a header count that changes inside the batch, bytes
    inside a value that look like a run of records right where a chunk begins, records that
    end well before the batch does, and a record count below the records present.  The walk
    of the whole batch decides.  (Offset deltas with gaps, a compacted batch, are among them
    too: its chunks count their records instead, see below.)"""
    rng = np.random.default_rng(2)
    more_headers = build_batch(dict(base_offset=0, last_offset_delta=3999),
                               records(4000, rng, headers=lambda i: 1 if i < 3000 else 3))
    fewer_headers = build_batch(dict(base_offset=0, last_offset_delta=3999),
                                records(4000, rng, headers=lambda i: 2 if i < 1000 else 0))
    gaps = build_batch(dict(base_offset=0, last_offset_delta=3 * 3999), records(4000, rng, od=lambda i: 3 * i))
    # a value that covers the start of chunk 1 and holds what looks like records 700, 701, ...
    fake = b"".join(bytes([2 * 9, 0]) + synth.varint(0) + synth.varint(700 + j) + b"\x01\x02\x03\x04\x05"
                    for j in range(60))
    plain = records(3000, rng)
    first = build_batch(dict(base_offset=0, last_offset_delta=2999), plain)
    st, cc, _, _ = oracle.decode_columns(first, offsets_of([first]))
    ends = 61 + cc["val_off"][:3000].astype(np.int64) + 100   # frame offsets of the records' ends
    at = int(np.searchsorted(ends, 21 + CHUNK - 300))          # the record that ends ~300 bytes before chunk 1
    plain[at + 1] = C.Record(0, 0, at + 1, at + 1, b"k", fake + bytes(200), [])
    mimic = build_batch(dict(base_offset=0, last_offset_delta=2999), plain)
    tail = bytearray(build_batch(dict(base_offset=0, last_offset_delta=1999), records(2000, rng)))
    tail += bytes(rng.integers(0, 256, 3 * CHUNK, dtype=np.uint8))      # bytes behind the records, inside the batch
    tail[8:12] = struct.pack(">i", len(tail) - 12)
    fewer = bytearray(build_batch(dict(base_offset=0, last_offset_delta=3999), records(4000, rng)))
    fewer[57:61] = struct.pack(">I", 2500)
    batches = [more_headers, fewer_headers, gaps, mimic, bytes(tail), bytes(fewer)]
    assert all(len(b) > 2 * CHUNK for b in batches)
    g, c, totals, *_ = decode(codec, batches)
    assert list(c["status"][:6]) == [C.OK] * 6
    assert totals.n_failed == 0 and totals.n_records == 4000 * 3 + 3000 + 2000 + 2500
    assert list(g["hdr_count"][:6]) == [3000 + 3 * 1000, 2000, 0, 0, 0, 0]


def patched(codec, batches):
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8).copy()
    off = offsets_of(batches)
    crc_patch(codec, blob, off)
    return blob, off


def walked_whole(codec, frame):
    """One decode of `frame` (patched()), held to the oracle: the batches whose chunks did not join up."""
    check_against_oracle(codec, *frame)
    return codec.last_walked_whole()


def test_compacted_batches_chunk_by_chunk(codec):
    """Batches that do not number their records 0, 1, 2 ... (what log compaction leaves: offset deltas
    with gaps, a last_offset_delta that is not the record count less one) cannot take a chunk's first
    record index from its offset delta: every chunk's wave counts the records that start in its chunk
    and adds up the counts of the chunks before it.  The chunks join up -- no batch is walked whole --
    and the columns are the sequential walk's, deltas that grow a byte at 64 and 8192 and all; whichever
    launch works on the chunks."""
    import os
    rng = np.random.default_rng(21)
    every_third = build_batch(dict(base_offset=0, last_offset_delta=synth.every_third_missing(9999)),
                              records(10000, rng, od=synth.every_third_missing, ts=synth.every_third_missing))
    tripled = build_batch(dict(base_offset=20000, last_offset_delta=3 * 3999),
                          records(4000, rng, od=lambda i: 3 * i, headers=lambda i: 2))
    # irregular gaps, and the batch's last records gone (last_offset_delta is the batch's as it was)
    kept = np.sort(rng.choice(30000, 6000, replace=False))
    irregular = build_batch(dict(base_offset=40000, last_offset_delta=31000),
                            records(6000, rng, od=lambda i: int(kept[i]), ts=lambda i: 7 * int(kept[i]), headers=lambda i: 1))
    # a numbered batch and small ones beside them
    plain = build_batch(dict(base_offset=80000, last_offset_delta=4999), records(5000, rng))
    small = [build_batch(dict(base_offset=90000 + 10 * i, last_offset_delta=9), records(10, rng)) for i in range(6)]
    batches = small[:3] + [every_third, plain, tripled] + small[3:] + [irregular]
    assert all(len(b) > 4 * CHUNK for b in (every_third, tripled, irregular))
    frame = patched(codec, batches)
    for beside in ("1", "0"):
        os.environ["NISSHI_CHUNKS_BESIDE"] = beside
        os.environ["NISSHI_ADAPTIVE"] = "0"       # every launch in every call
        try:
            fresh = hip.Codec(0)
        finally:
            del os.environ["NISSHI_CHUNKS_BESIDE"], os.environ["NISSHI_ADAPTIVE"]
        try:
            assert walked_whole(fresh, frame) == 0
            assert walked_whole(fresh, frame) == 0      # (the counts an earlier call left are not this call's)
        finally:
            fresh.close()


def test_what_compaction_itself_leaves(codec):
    """A partition of large batches through the library's own compaction (nisshi_hip_compact: the last
    record of every key stays), and the batches it rewrote decoded again: gaps wherever a key came again
    later, the batch's last_offset_delta as it was.  Their chunks join up."""
    rng = np.random.default_rng(23)
    batches, base = [], 0
    for b in range(4):
        n = 7000
        keys = rng.integers(0, 14000, n)
        batches.append(build_batch(dict(base_offset=base, last_offset_delta=n - 1),
                                   records(n, rng, value=lambda i: 100, key=lambda i: b"key-%08d" % keys[i])))
        base += n
    blob, off = patched(codec, batches)
    pieces, removed, action, status, totals = codec.compact(blob.tobytes(), off)
    st, want_removed, want_action, want_pieces = oracle.compact_partition([blob[int(off[i]):int(off[i + 1])].tobytes()
                                                                          for i in range(4)])
    assert st == C.OK and pieces == want_pieces and list(action) == [1, 1, 1, 1]
    assert all(1000 < r < 6000 for r in removed)
    assert all(len(p) > 2 * CHUNK + 4096 for p in pieces)
    fresh = hip.Codec(0)
    try:
        assert walked_whole(fresh, patched(codec, pieces)) == 0
    finally:
        fresh.close()


def test_compacted_batches_through_the_other_entry_points(codec):
    """A compacted batch of sixteen chunks among seven thousand small batches (a call of that many has
    its chunks worked on by a launch of their own, from the second call on: the first leaves word of
    them); the same frame through the host-buffer entry point, and through the one that allocates
    nothing.  All against the checker."""
    from util import assert_same
    rng = np.random.default_rng(24)
    big = build_batch(dict(base_offset=0, last_offset_delta=synth.every_third_missing(9999) + 5),
                      records(10000, rng, od=synth.every_third_missing, ts=lambda i: 2 * i, headers=lambda i: 1))
    small = build_batch(dict(base_offset=50000, last_offset_delta=4), records(5, rng, od=lambda i: i))
    assert len(big) > 16 * CHUNK
    blob, off = patched(codec, [small] * 3500 + [big] + [small] * 3500)
    st, want, nr, nh = oracle.decode_columns(blob, off)
    assert st == C.OK
    fresh = hip.Codec(0)
    try:
        d_blob, d_off = fresh.to_device(blob, off)
        n_b = len(off) - 1
        for call in range(3):
            cols = hip.DeviceColumns(n_b, nr, nh, fresh.device)
            fresh.profile(True)
            totals = fresh.decode_dev(d_blob, len(blob), d_off, n_b, cols, static=call == 2)
            launched = "chunk_kernel" in fresh.kernel_times()
            fresh.profile(False)
            assert launched == (call == 1)      # (the entry point that allocates nothing records no events)
            assert (totals.n_records, totals.n_headers, totals.n_failed) == (nr, nh, 0)
            assert_same(cols.to_host(nr, nh), want, n_b)
        got, totals = fresh.decode(blob.tobytes(), off)
        assert (totals.n_records, totals.n_headers, totals.n_failed) == (nr, nh, 0)
        assert_same(got, want, n_b)
    finally:
        fresh.close()


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_compacted_batches_of_many_makes(codec, seed):
    """Drawn at random: how many records, which of them a compaction took (runs of survivors and of gaps of
    drawn lengths), value and key sizes (one size a batch, or a few sizes in stretches), header counts (none, or
    the same for every record), timestamps that jump, a last_offset_delta beyond the last record.  Whatever
    the chunks make of a batch -- joined, or walked whole -- the columns are the checker's."""
    rng = np.random.default_rng(seed)
    batches, base = [], 0
    for b in range(14):
        n = int(rng.integers(1500, 9000))
        # survivors: runs of kept records between gaps
        kept, at = [], 0
        while len(kept) < n:
            run = int(rng.integers(1, 400))
            kept.extend(range(at, at + run))
            at += run + int(rng.integers(0, 3 if b % 3 else 300))
        kept = kept[:n]
        sizes = [int(rng.integers(20, 300)) for _ in range(1 if b % 2 else 3)]
        stretch = int(rng.integers(200, 3000))
        hc = int(rng.integers(0, 3)) if b % 4 else 0
        klen = int(rng.integers(0, 30))
        jump = int(rng.integers(1, 5000))
        # (now and then what no guess holds for: a header count, or sizes, that change from record to record)
        odd_hc, odd_size = b % 7 == 5, b % 7 == 6
        recs = records(n, rng, od=lambda i: kept[i], ts=lambda i: jump * kept[i] - (i % 7),
                       value=lambda i: sizes[(i // stretch) % len(sizes)] + (i % 5 if odd_size else 0),
                       headers=lambda i: hc + (i % 3 == 0 if odd_hc else 0),
                       key=lambda i: (b"k" * klen) if klen else None)
        batches.append(build_batch(dict(base_offset=base, last_offset_delta=kept[-1] + int(rng.integers(0, 50))), recs))
        base += kept[-1] + 100
    frame = patched(codec, batches)
    for beside in ("1", "0"):
        os.environ["NISSHI_CHUNKS_BESIDE"] = beside
        os.environ["NISSHI_ADAPTIVE"] = "0"
        try:
            fresh = hip.Codec(0)
        finally:
            del os.environ["NISSHI_CHUNKS_BESIDE"], os.environ["NISSHI_ADAPTIVE"]
        try:
            whole = walked_whole(fresh, frame)
            assert whole <= len(batches)
            if beside == "1":
                assert whole <= 5       # (batches of one size join up; a change of size inside a chunk is found as well)
        finally:
            fresh.close()


def test_compacted_batches_of_records_of_every_size(codec):
    """Sizes that change from record to record (values and keys of drawn lengths, records on both sides of
    the 64 bytes where the length takes a second byte) in batches that do not number their records: the
    strides of the count find nothing to hold on to and the chunk's records are counted from length to
    length, by 64 chains of hops side by side where the records are long enough to be told by their first
    bytes, one after the other where not.  The chunks join up all the same, whichever launch works on them."""
    import os
    rng = np.random.default_rng(25)
    gaps = lambda i: 2 * i + 1
    v_len, k_len = rng.integers(0, 200, 12000), rng.integers(0, 40, 12000)
    values = build_batch(dict(base_offset=0, last_offset_delta=gaps(11999)),
                         records(12000, rng, od=gaps, value=lambda i: 20 + int(v_len[i])))
    keys = build_batch(dict(base_offset=30000, last_offset_delta=synth.every_third_missing(7999) + 9),
                       records(8000, rng, od=synth.every_third_missing, ts=lambda i: 5 * i, headers=lambda i: 1,
                               value=lambda i: int(v_len[i]) // 2, key=lambda i: b"k" * int(k_len[i]) if k_len[i] else None))
    # (one size for a stretch, then every size: the strides hand over to the hops inside a chunk)
    mixed = build_batch(dict(base_offset=60000, last_offset_delta=gaps(5999)),
                        records(6000, rng, od=gaps, value=lambda i: 100 if i % 1500 < 900 else 60 + int(v_len[i]) % 90))
    # (records of under 64 bytes, whose lengths take one byte: no chain of hops finds a start to begin at beside
    # the first, and the records are counted one after the other)
    short = build_batch(dict(base_offset=80000, last_offset_delta=gaps(15999)),
                        records(16000, rng, od=gaps, value=lambda i: 18 + int(v_len[i % 12000]) % 20, key=lambda i: None))
    small = [build_batch(dict(base_offset=190000 + 10 * i, last_offset_delta=9), records(10, rng)) for i in range(4)]
    assert all(len(b) > 6 * CHUNK for b in (values, keys, mixed, short))
    frame = patched(codec, small[:2] + [values, keys] + small[2:] + [mixed, short])
    for beside in ("1", "0"):
        os.environ["NISSHI_CHUNKS_BESIDE"] = beside
        os.environ["NISSHI_ADAPTIVE"] = "0"
        try:
            fresh = hip.Codec(0)
        finally:
            del os.environ["NISSHI_CHUNKS_BESIDE"], os.environ["NISSHI_ADAPTIVE"]
        try:
            assert walked_whole(fresh, frame) == 0
            assert walked_whole(fresh, frame) == 0
        finally:
            fresh.close()


@pytest.mark.parametrize("short", [False, True])
def test_compacted_batches_of_every_size_by_the_hundred(codec, short):
    """A hundred batches of 5000 records with every third gone and sizes drawn a record (values of 32 to 96
    bytes and two headers; or, `short`, records of 21 to 53 bytes without headers, whose lengths take one
    byte), every column held to the checker; the chunks of every batch join up."""
    kind = dict(key_len=4, val_range=(8, 40), n_headers=0) if short else {}
    blob, off, nr, nh = synth.make_var_batches(100, 5000, synth.SEED_BASE + 11, delta=synth.every_third_missing, **kind)
    crc_patch(codec, blob, off)
    fresh = hip.Codec(0)
    try:
        g, c, totals, *_ = check_against_oracle(fresh, blob, off)
        assert (totals.n_records, totals.n_headers, totals.n_failed, totals.n_exact) == (nr, nh, 0, 0)
        assert fresh.last_walked_whole() == 0
    finally:
        fresh.close()


def test_compacted_batches_that_the_count_cannot_follow(codec):
    """The count looks at a few places a chunk and takes records of one size to lie between them, or goes
    from record to record up to a few thousand a chunk.  Where neither holds -- one record of another size
    and then the old size again may slip through the strides, bytes behind the records, records of a dozen
    bytes and changing sizes -- the chunks do not join up (or are not counted at all) and the batch is
    walked whole: same columns, and the context says how many such batches there were."""
    rng = np.random.default_rng(22)
    gaps = lambda i: 2 * i + 1
    one_off = build_batch(dict(base_offset=10000, last_offset_delta=gaps(3999)),
                          records(4000, rng, od=gaps, value=lambda i: 101 if i in (700, 2500) else 100))
    tail = bytearray(build_batch(dict(base_offset=20000, last_offset_delta=gaps(1999)), records(2000, rng, od=gaps)))
    tail += bytes(rng.integers(0, 256, 3 * CHUNK, dtype=np.uint8))
    tail[8:12] = struct.pack(">i", len(tail) - 12)
    tiny = build_batch(dict(base_offset=25000, last_offset_delta=gaps(29999)),
                       records(30000, rng, od=gaps, value=lambda i: i % 3, key=lambda i: None))
    fine = build_batch(dict(base_offset=90000, last_offset_delta=gaps(3999)), records(4000, rng, od=gaps))
    assert len(tiny) > 4 * CHUNK
    frame, fines = patched(codec, [one_off, bytes(tail), tiny, fine]), patched(codec, [fine, fine])
    fresh = hip.Codec(0)
    try:
        assert 2 <= walked_whole(fresh, frame) <= 3
        assert walked_whole(fresh, fines) == 0
    finally:
        fresh.close()


def test_c5_after_compaction_vs_oracle(codec):
    """The ragged workload with every third record of every batch gone, through both schedules (chunks
    beside the batches; a launch of their own), twice each."""
    import os
    shape, blob, off, nr, nh, nb = __import__("tansu_amd.shard", fromlist=["x"]).make_shard("c5g", 0.012, 0)
    crc_patch(codec, blob, off)
    for beside in ("1", "0"):
        os.environ["NISSHI_CHUNKS_BESIDE"] = beside
        try:
            other = hip.Codec(0)
        finally:
            del os.environ["NISSHI_CHUNKS_BESIDE"]
        try:
            for _ in range(2):
                g, c, totals, *_ = check_against_oracle(other, blob, off)
                assert (totals.n_records, totals.n_failed, totals.n_exact) == (nr, 0, 0)
            assert other.last_walked_whole() == 0
        finally:
            other.close()


def test_chunks_with_records_that_do_not_decode(codec):
    """A record whose length disagrees with its fields in chunk 2 (the reference takes
    the fields' word for it and goes on, record.rs:254-277), a record cut short by the
    end of the data in the last chunk, a count beyond the records: the batch's outcome is
    the sequential walk's."""
    rng = np.random.default_rng(3)
    good = build_batch(dict(base_offset=0, last_offset_delta=3999), records(4000, rng))
    st, cc, _, _ = oracle.decode_columns(good, offsets_of([good]))
    wrong = bytearray(good)
    # record 1500's length varint: two bytes before its attributes byte; its key starts 3 + 2 + 2 bytes further on
    start = 61 + int(cc["key_off"][1500]) - 1 - 2 - 2 - 1 - 2
    assert wrong[start + 2] == 0 and start > 21 + 2 * CHUNK
    wrong[start] ^= 0x02
    cut = bytearray(good)
    cut[8:12] = struct.pack(">i", len(cut) - 12 - 40)
    del cut[-40:]
    over = bytearray(good)
    over[57:61] = struct.pack(">I", 4001)
    batches = [good, bytes(wrong), bytes(cut), bytes(over), good]
    g, c, totals, *_ = decode(codec, batches)
    assert [int(s) == C.OK for s in c["status"][:5]] == [True, True, False, False, True]
    assert totals.n_failed == 2 and totals.n_exact >= 3


def test_compacted_batches_of_every_size_with_records_that_do_not_decode(codec):
    """The same damage in batches that do not number their records and whose sizes change from record
    to record, where a chunk's wave goes from length to length to count its records: a length that is
    off by one in chunk 2, a length whose second byte goes on, an attributes byte with a bit set, the
    data cut short, a count beyond the records, random bytes for a whole chunk.  Whatever the count
    makes of them, the batch's outcome is the sequential walk's."""
    rng = np.random.default_rng(35)
    gaps = lambda i: 2 * i + 1
    v_len = rng.integers(0, 200, 6000)
    good = build_batch(dict(base_offset=0, last_offset_delta=gaps(5999)),
                       records(6000, rng, od=gaps, value=lambda i: 20 + int(v_len[i])))
    assert len(good) > 8 * CHUNK
    st, cc, _, _ = oracle.decode_columns(good, offsets_of([good]))
    # the first byte of the length of a record of a hundred bytes and more behind record 2500, from where its key
    # starts: length (2), attributes, two deltas of two bytes, the key's length
    which = 2500 + int(np.argmax(v_len[2500:] >= 80))
    start = 61 + int(cc["key_off"][which]) - 1 - 2 - 2 - 1 - 2
    assert good[start] & 0x80 and good[start + 2] == 0 and start > 21 + 2 * CHUNK

    def damaged(at, how):
        b = bytearray(good)
        b[at] = how(b[at])
        return bytes(b)

    off_by_one = damaged(start, lambda v: v ^ 0x02)
    goes_on = damaged(start + 1, lambda v: v | 0x80)
    attributes = damaged(start + 2, lambda v: 0x08)
    cut = bytearray(good)
    cut[8:12] = struct.pack(">i", len(cut) - 12 - 40)
    del cut[-40:]
    over = bytearray(good)
    over[57:61] = struct.pack(">I", 6001)
    noise = bytearray(good)
    noise[61 + 3 * CHUNK:61 + 4 * CHUNK] = bytes(rng.integers(0, 256, CHUNK, dtype=np.uint8))
    batches = [good, off_by_one, goes_on, attributes, bytes(cut), bytes(over), bytes(noise), good]
    g, c, totals, *_ = decode(codec, batches)
    ok = [int(s) == C.OK for s in c["status"][:8]]
    assert ok[0] and ok[7] and not ok[4] and not ok[5] and not ok[6]
    assert totals.n_failed == 8 - sum(ok)


def test_few_and_many_batches_give_the_same_columns(codec):
    """A call of few batches takes its batches of 128 KiB and more chunk by chunk, a
    call of many only those of 512 KiB and more: the same batch decodes to the same
    columns in both (here through the context's switch for the number of batches)."""
    import os
    shape, counts, seed = synth.config("c5", 0.01)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    crc_patch(codec, blob, off)
    g, c, totals, *_ = check_against_oracle(codec, blob, off)
    os.environ["NISSHI_SEG_FEW_BATCHES"] = "0"
    try:
        other = hip.Codec(0)
        g2, c2, totals2, *_ = check_against_oracle(other, blob, off)
        other.close()
    finally:
        del os.environ["NISSHI_SEG_FEW_BATCHES"]
    # (slots past the records and headers decoded are never written)
    header_level = {f for _, fields in C.HEADER_GROUPS for f in fields}
    for name in g:
        n = totals.n_headers if name in header_level else len(g[name])
        if name == "enc_len":   # which walk took a batch decides whether its size was noted, not what it is
            both = (g[name] != 0) & (g2[name] != 0)
            assert both.any() and (g[name][both] == g2[name][both]).all()
            continue
        assert (g[name][:n] == g2[name][:n]).all(), name
    assert totals.n_records == totals2.n_records == nr
    assert totals.n_headers == totals2.n_headers == nh


def test_batches_of_many_records_are_written_in_pieces(codec):
    """The encoder writes a batch of 2048 records and more 1024 records a wave, each
    piece adding its share to the batch's CRC; a long value inside a piece is copied
    apart and leaves the CRC to the pass over the output.  Output against the oracle's
    and against the (canonical) input, CRC field included."""
    import torch
    rng = np.random.default_rng(4)
    plain = build_batch(dict(base_offset=0, last_offset_delta=4999), records(5000, rng, headers=lambda i: i % 3))
    long_value = build_batch(dict(base_offset=0, last_offset_delta=2999),
                             records(3000, rng, value=lambda i: 40000 if i == 1700 else 700 if i == 2100 else 50))
    edge = build_batch(dict(base_offset=0, last_offset_delta=2047), records(2048, rng, value=lambda i: i % 90))
    small = build_batch(dict(base_offset=0, last_offset_delta=2046), records(2047, rng))
    batches = [plain, long_value, edge, small, plain]
    g, c, totals, cols, d_blob, d_off = decode(codec, batches)
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8)
    n_b = len(batches)
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, n_b, d_blob, cols.t["data_off"], out, len(blob), out_off)
    got = out[:t.out_bytes].cpu().numpy().tobytes()
    st, want, _ = oracle.encode_columns(c, n_b, d_blob.cpu().numpy()[:len(blob)], c["data_off"])
    assert st == C.OK and got == want
    # the input but for the CRC fields, which the test's batches carry as the decoder computed them
    assert got == d_blob.cpu().numpy()[:len(blob)].tobytes()


def test_scheduling_switches_give_the_same_columns(codec):
    """Which launch works on the chunks (beside the batches in the fused kernel's
    launch, or in a launch of their own before it) and in which order the batches
    start (those above the mean size first, or as they lie) changes no column: a
    call with batches that join, one that does not, one of large records and small
    ones in between, through every setting of the context's switches."""
    import os
    rng = np.random.default_rng(11)
    joining = build_batch(dict(base_offset=0, last_offset_delta=5999), records(6000, rng, headers=lambda i: 1))
    gaps = build_batch(dict(base_offset=6000, last_offset_delta=11999), records(4000, rng, od=lambda i: 3 * i))
    large = build_batch(dict(base_offset=20000, last_offset_delta=5),
                        records(6, rng, value=lambda i: 150_000, key=lambda i: None))
    small = [build_batch(dict(base_offset=30000 + 10 * i, last_offset_delta=9), records(10, rng, headers=lambda i: i % 3))
             for i in range(20)]
    batches = small[:7] + [joining] + small[7:12] + [gaps, large] + small[12:]
    g, c, totals, *_ = decode(codec, batches)
    assert totals.n_failed == 0
    header_level = {f for _, fields in C.HEADER_GROUPS for f in fields}
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8).copy()
    off = offsets_of(batches)
    crc_patch(codec, blob, off)
    for beside, big_first in ((0, 0), (0, 1), (1, 0)):
        os.environ["NISSHI_CHUNKS_BESIDE"] = str(beside)
        os.environ["NISSHI_BIG_FIRST"] = str(big_first)
        try:
            other = hip.Codec(0)
            g2, c2, totals2, *_ = check_against_oracle(other, blob, off)
            other.close()
        finally:
            del os.environ["NISSHI_CHUNKS_BESIDE"], os.environ["NISSHI_BIG_FIRST"]
        assert (totals2.n_records, totals2.n_headers) == (totals.n_records, totals.n_headers)
        for name in g:
            n = totals.n_headers if name in header_level else len(g[name])
            assert (g[name][:n] == g2[name][:n]).all(), (name, beside, big_first)


def test_a_shape_guess_that_does_not_fit(codec):
    """The walking waves start with the shape of the first record of the first batch
    of their block of 256 batches in hand.  Here that batch is unlike the others of
    its block (another shape, no records at all, a first record that does not
    decode, a compressed batch): the guess is tested against records like any shape
    and costs nothing but the learning.  600 batches, so three blocks."""
    rng = np.random.default_rng(12)

    def plain(i):
        return build_batch(dict(base_offset=100 * i, last_offset_delta=69), records(70, rng, headers=lambda j: 2))

    firsts = [
        build_batch(dict(base_offset=0, last_offset_delta=9), records(10, rng, value=lambda j: 7, key=lambda j: None)),
        build_batch(dict(base_offset=25600, last_offset_delta=0), []),
        None,
    ]
    damaged = bytearray(plain(512))
    damaged[61 + 3] = 0xff  # the first record's timestamp delta runs on into the offset delta: it no longer decodes as its length says
    damaged[61 + 4] = 0xff
    firsts[2] = bytes(damaged)
    batches = []
    for i in range(600):
        batches.append(firsts[i // 256] if i % 256 == 0 else plain(i))
    g, c, totals, *_ = decode(codec, batches)
    assert totals.n_failed == int((c["status"] != C.OK).sum())
    assert int(c["status"][1]) == C.OK and int(c["status"][257]) == C.OK
