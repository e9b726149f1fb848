"""The CPU restatement of the fetch seam of the SQL engines (nisshi-storage/src/
pg.rs:1875-2052, lite.rs:2765-2980).  The reference has no unit test of the seam
itself (it needs a database): the restatement is checked against a Python model
of the reference's loop, and its pieces against the reference's known answers
(the builder's record length, record.rs:478-523; the builder defaults,
inflated.rs:311-329)."""
import numpy as np
import pytest

import oracle
from fetch_cases import CASES, NOW, make_rows, model_fetch
from tansu_amd import columns as C
from util import offsets_of


def decode_all(batches):
    blob = b"".join(batches)
    st, cols, nr, nh = oracle.decode_columns(blob, offsets_of(batches), True)
    assert st == C.OK and (cols["status"] == C.OK).all()
    return cols, [C.records_of_batch(cols, b, blob) for b in range(len(batches))]


def test_no_rows_gives_the_default_batch():
    # pg.rs:2050-2052 with inflated.rs:311-329
    rows, payload = make_rows(0)
    st, batches = oracle.fetch(rows, payload, NOW)
    assert st == C.OK and len(batches) == 1 and len(batches[0]) == 61
    cols, recs = decode_all(batches)
    want = dict(base_offset=0, partition_leader_epoch=-1, magic=2, attributes=0, last_offset_delta=0,
                base_timestamp=NOW, max_timestamp=NOW, producer_id=-1, producer_epoch=0, base_sequence=0,
                record_count=0, batch_length=49)
    for k, v in want.items():
        assert int(cols[k][0]) == v, k
    assert batches == model_fetch(rows, payload)


def test_record_length_is_the_builders():
    # record.rs:478-523: value "def", no key, deltas 0 -> length 9, bytes [18,0,0,0,1,6,100,101,102,0]
    rows = dict(offset=[5], attributes=[0], timestamp=[77], producer_id=[-1], producer_epoch=[-1],
                completed=[1], key_off=[0], key_len=[-1], val_off=[0], val_len=[3], hdr_base=[0, 0],
                hdr=np.zeros((0, 4), dtype=np.uint32))
    st, batches = oracle.fetch(rows, b"def", NOW)
    assert st == C.OK and len(batches) == 1
    assert batches[0][61:] == bytes([18, 0, 0, 0, 1, 6, 100, 101, 102, 0])
    cols, recs = decode_all(batches)
    assert int(cols["base_offset"][0]) == 5 and int(cols["base_timestamp"][0]) == 77
    assert int(cols["max_timestamp"][0]) == NOW and int(cols["base_sequence"][0]) == 0
    assert recs[0][0].length == 9 and recs[0][0].value == b"def" and recs[0][0].key is None


@pytest.mark.parametrize("name", sorted(CASES))
def test_matches_the_model(name):
    kw = dict(CASES[name])
    rows, payload = make_rows(kw.pop("n"), seed=len(name), **kw)
    want = model_fetch(rows, payload)
    st, got = oracle.fetch(rows, payload, NOW)
    if want is None:
        assert st == C.TRY_FROM_INT and got == []
        return
    assert st == C.OK
    assert got == want


def test_open_transaction_ends_the_fetch():
    # pg.rs:1965-1971: rows 0..2 complete, row 3 open and not adjacent
    rows, payload = make_rows(6, seed=3, gap_rate=0.0)
    rows["completed"][:] = [1, 1, 1, 0, 1, 1]
    st, got = oracle.fetch(rows, payload, NOW)
    cols, recs = decode_all(got)
    assert [len(r) for r in recs] == [6]          # adjacent offsets: the open row continues
    rows["offset"][3:] += 1
    st, got = oracle.fetch(rows, payload, NOW)
    cols, recs = decode_all(got)
    assert [len(r) for r in recs] == [3] and int(cols["last_offset_delta"][0]) == 2
    # the stop row may still open a batch, which stays empty (pg.rs:1933-1971)
    rows["producer_id"][3:] = 99
    st, got = oracle.fetch(rows, payload, NOW)
    cols, recs = decode_all(got)
    assert [len(r) for r in recs] == [3, 0]
    assert int(cols["base_offset"][1]) == int(rows["offset"][3]) and int(cols["producer_id"][1]) == 99
    assert got == model_fetch(rows, payload)


def test_capacity():
    rows, payload = make_rows(30, seed=1, change_every=3)
    st, got = oracle.fetch(rows, payload, NOW, batch_cap=4)
    assert st == C.CAPACITY
