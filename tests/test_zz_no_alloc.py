"""nisshi_hip_decode_dev_static (the decode entry point that allocates nothing)
against nisshi_hip_decode_dev and against the CPU checker (oracle/), through the
C ABI: the same batches must give the same columns, bit for bit, on two calls in
a row, the second with a smaller input (the static workspace keeps what the
first call left in it)."""
import ctypes

import numpy as np
import pytest

import oracle
from tansu_amd import columns as C, hip, synth
from util import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def mixed_blob(scale):
    """Batches of three kinds in one blob: small records with headers, large
    batches (chunk-parallel CRC, several waves a batch), ragged counts."""
    parts = [synth.make_batches(synth.C2_SHAPE, np.full(max(2, int(24 * scale)), 1000), synth.SEED_BASE + 71),
             synth.make_batches(synth.Shape(key_len=-1, val_len=1 << 18, n_headers=1), np.full(max(1, int(3 * scale)), 3),
                                synth.SEED_BASE + 72),
             synth.make_batches(synth.C1_SHAPE, np.array([1, 7, 640, 5000, 64][:max(2, int(5 * scale))]),
                                synth.SEED_BASE + 73)]
    blob = np.concatenate([p[0] for p in parts])
    off, base = [np.zeros(1, dtype=np.uint64)], 0
    for b, o, _, _ in parts:
        off.append(np.asarray(o[1:], dtype=np.uint64) + np.uint64(base))
        base += len(b)
    return blob, np.concatenate(off)


def both(codec, blob, off):
    n_b = len(off) - 1
    d_blob, d_off = codec.to_device(blob, off)
    rec_cap = codec.count_dev(d_blob, len(blob), d_off, n_b)
    hdr_cap = len(blob) // 2 + 4096 * n_b + 1
    got = []
    for static in (False, True):
        cols = hip.DeviceColumns(n_b, rec_cap, hdr_cap, codec.device)
        for t in cols.t.values():
            t.zero_()
        totals = codec.decode_dev(d_blob, len(blob), d_off, n_b, cols, static=static)
        got.append((cols.to_host(totals.n_records, min(totals.n_headers, hdr_cap)), totals))
    (a, ta), (b, tb) = got
    assert (ta.n_records, ta.n_failed, ta.n_exact) == (tb.n_records, tb.n_failed, tb.n_exact)
    assert int(a["hdr_count"][:n_b].sum()) == int(b["hdr_count"][:n_b].sum())
    assert ta.n_failed == 0 and (a["crc_computed"][:n_b] == b["crc_computed"][:n_b]).all()
    assert_same(b, a, n_b)   # every batch and record field, headers through each record's slots
    # and both against the CPU restatement of the reference's decode
    st, want, nr, nh = oracle.decode_columns(blob, off)
    assert st == C.OK and (tb.n_records, int(b["hdr_count"][:n_b].sum())) == (nr, nh)
    assert_same(b, want, n_b)
    assert_same(a, want, n_b)
    return ta


def test_static_entry_point_matches_on_two_calls_in_a_row(codec):
    blob, off = mixed_blob(1.0)
    t1 = both(codec, blob, off)
    small, small_off = mixed_blob(0.4)
    assert len(small) < len(blob) and len(small_off) < len(off)
    t2 = both(codec, small, small_off)
    assert 0 < t2.n_records < t1.n_records


def test_static_entry_point_refuses_what_exceeds_its_workspace(codec):
    import torch
    lib = codec.lib
    d = torch.zeros(64, dtype=torch.uint8, device=codec.device)
    o = torch.zeros(4, dtype=torch.int64, device=codec.device)
    cols = hip.DeviceColumns(1, 1, 1, codec.device)
    totals = hip.Totals()
    st = lib.nisshi_hip_decode_dev_static(codec.ctx, d.data_ptr(), 64, o.data_ptr(), 16385, 1, ctypes.byref(cols.c),
                                          1, 1, ctypes.byref(totals), None)
    assert st == C.CAPACITY
    assert b"16384" in lib.nisshi_hip_last_error(codec.ctx)
    st = lib.nisshi_hip_decode_dev_static(codec.ctx, d.data_ptr(), (512 << 20) + 1, o.data_ptr(), 1, 1,
                                          ctypes.byref(cols.c), 1, 1, ctypes.byref(totals), None)
    assert st == C.CAPACITY
