"""The CPU restatement of log compaction (inflated::Batch::compact and keys,
nisshi-sans-io/src/record/inflated.rs:213-275; its caller nisshi-storage/src/
slate/storage.rs:283-312), pinned by the reference's own tests of compact
(inflated.rs:664-790) and checked against a Python model of the partition walk."""
import numpy as np
import pytest

import oracle
from compact_cases import CASES, keyed, make_partition, model_compact
from tansu_amd import columns as C
from util import BATCH_FIELDS, build_batch, offsets_of


def decode(batch):
    st, cols, nr, nh = oracle.decode_columns(batch, offsets_of([batch]), True)
    assert st == C.OK and cols["status"][0] == C.OK
    hdr = {k: int(cols[k][0]) for k in ("base_offset", "partition_leader_epoch", "magic", "attributes",
                                        "last_offset_delta", "base_timestamp", "max_timestamp", "producer_id",
                                        "producer_epoch", "base_sequence")}
    return hdr, C.records_of_batch(cols, 0, batch)


def compact_one(batch, head_keys=()):
    """batch.compact(&head): the head as a newer batch holding those keys."""
    batches = [batch]
    if head_keys:
        batches.append(build_batch(dict(base_offset=100), [keyed(i, k, b"x") for i, k in enumerate(head_keys)]))
    st, removed, action, pieces = oracle.compact_partition(batches)
    assert st == C.OK
    return int(removed[0]), int(action[0]), pieces[0]


def test_compact_noop_without_duplicates():
    # inflated.rs:664-680
    batch = build_batch(dict(last_offset_delta=1), [keyed(0, b"k1", b"v1"), keyed(1, b"k2", b"v2")])
    assert compact_one(batch) == (0, oracle.UNCHANGED, None)


def test_compact_within_batch_keeps_last_occurrence_per_key():
    # inflated.rs:682-709
    batch = build_batch(dict(base_offset=5, last_offset_delta=2),
                        [keyed(0, b"k1", b"v1"), keyed(1, b"k2", b"v2"), keyed(2, b"k1", b"v3")])
    removed, action, piece = compact_one(batch)
    assert (removed, action) == (1, oracle.REWRITTEN)
    hdr, recs = decode(piece)
    assert len(recs) == 2
    assert recs[0].offset_delta == 1 and recs[0].key == b"k2"
    assert recs[1].offset_delta == 2 and recs[1].value == b"v3"
    assert hdr["base_offset"] == 5 and hdr["last_offset_delta"] == 2


def test_compact_drops_keys_in_head():
    # inflated.rs:711-728
    batch = build_batch(dict(last_offset_delta=1), [keyed(0, b"k1", b"v1"), keyed(1, b"k2", b"v2")])
    assert compact_one(batch, [b"k1", b"k2"]) == (2, oracle.DELETED, None)


def test_compact_retains_keyless_records():
    # inflated.rs:730-757
    batch = build_batch(dict(last_offset_delta=2),
                        [C.Record(0, 0, 0, 0, None, b"v0"), keyed(1, b"k1", b"v1"), keyed(2, b"k1", b"v2")])
    removed, action, piece = compact_one(batch)
    assert (removed, action) == (1, oracle.REWRITTEN)
    _, recs = decode(piece)
    assert [(r.offset_delta, r.key) for r in recs] == [(0, None), (2, b"k1")]


def test_compact_retains_keyless_when_all_keyed_records_drop():
    # inflated.rs:759-784
    batch = build_batch(dict(last_offset_delta=1), [C.Record(0, 0, 0, 0, None, b"v0"), keyed(1, b"k1", b"v1")])
    removed, action, piece = compact_one(batch, [b"k1"])
    assert (removed, action) == (1, oracle.REWRITTEN)
    _, recs = decode(piece)
    assert [(r.offset_delta, r.key) for r in recs] == [(0, None)]


@pytest.mark.parametrize("name", sorted(CASES))
def test_matches_the_model(name):
    batches = make_partition(seed=len(name), **CASES[name])
    want = model_compact(batches, decode)
    st, removed, action, pieces = oracle.compact_partition(batches)
    assert st == C.OK
    assert (list(removed), list(action), pieces) == want
    if name != "all_keyless":
        assert sum(want[0]) > 0


def test_failed_batch_fails_the_compaction():
    batches = make_partition(3, 10, 4)
    batches[1] = batches[1][:40]
    st, *_ = oracle.compact_partition(batches)
    assert st != C.OK
