"""Parity of the log compaction operator (nisshi-sans-io/src/record/inflated.rs:
213-275, nisshi-storage/src/slate/storage.rs:283-312) with its CPU restatement,
through the C ABI.  Bit-exact: counts, actions and the rewritten batches' bytes."""
import numpy as np
import pytest

import oracle
from compact_cases import CASES, keyed, make_partition
from tansu_amd import columns as C, hip
from test_gpu_parity import gpu_decode
from util import build_batch, offsets_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def check_host(codec, batches):
    st, removed, action, pieces = oracle.compact_partition(batches)
    assert st == C.OK
    got, g_removed, g_action, g_status, totals = codec.compact(b"".join(batches), offsets_of(batches))
    assert list(g_removed) == list(removed)
    assert list(g_action) == list(action)
    assert got == pieces
    assert [int(s) for s in g_status] == [C.OK if a == 1 else hip.NONE for a in action]
    return removed


@pytest.mark.parametrize("name", sorted(CASES))
def test_cases_host(codec, name):
    removed = check_host(codec, make_partition(seed=len(name), **CASES[name]))
    if name != "all_keyless":
        assert removed.sum() > 0


def test_reference_cases(codec):
    # inflated.rs:664-790, the head as a newer batch holding those keys
    newer = build_batch(dict(base_offset=100), [keyed(0, b"k1", b"x"), keyed(1, b"k2", b"x")])
    for recs, head in [
            ([keyed(0, b"k1", b"v1"), keyed(1, b"k2", b"v2")], False),
            ([keyed(0, b"k1", b"v1"), keyed(1, b"k2", b"v2"), keyed(2, b"k1", b"v3")], False),
            ([keyed(0, b"k1", b"v1"), keyed(1, b"k2", b"v2")], True),
            ([C.Record(0, 0, 0, 0, None, b"v0"), keyed(1, b"k1", b"v1"), keyed(2, b"k1", b"v2")], False),
            ([C.Record(0, 0, 0, 0, None, b"v0"), keyed(1, b"k1", b"v1")], True)]:
        batch = build_batch(dict(base_offset=5, last_offset_delta=len(recs) - 1), recs)
        check_host(codec, [batch] + ([newer] if head else []))


def test_device_form_and_records(codec):
    """The compacted columns themselves: kept records in order, lengths rebuilt only in
    batches that lost records; encode of the compacted columns gives the rewritten batches."""
    import torch
    batches = make_partition(10, 120, 30, seed=4, headers=True)
    # a stored length that is not the body size survives in a batch that loses nothing
    blob, off = b"".join(batches), offsets_of(batches)
    n_b = len(batches)
    cols, totals, d_blob, d_off = gpu_decode(codec, blob, off, hdr_cap=len(blob) // 2 + 4096 * n_b + 1)
    out, removed, action, t = codec.compact_dev(d_blob, cols, n_b, totals.n_records)
    st, want_removed, want_action, pieces = oracle.compact_partition(batches)
    assert removed.cpu().numpy()[:n_b].tolist() == list(want_removed)
    assert action.cpu().numpy()[:n_b].tolist() == list(want_action)
    g = out.to_host()
    kept = 0
    for b in range(n_b):
        if want_action[b] == 1:
            stp, pc, _, _ = oracle.decode_columns(pieces[b], offsets_of([pieces[b]]), True)
            n = int(pc["record_count"][0])
            assert int(g["record_count"][b]) == n
            r0 = int(g["rec_base"][b])
            for name in ("length", "rattributes", "timestamp_delta", "offset_delta", "key_len", "val_len", "rhdr_count"):
                assert (g[name][r0:r0 + n] == pc[name][:n]).all(), (b, name)
        kept += int(g["record_count"][b])
    assert t.n_records == kept
    cap = len(blob) + 64
    d_out = torch.zeros(cap + 16, dtype=torch.uint8, device=codec.device)
    d_out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
    te = codec.encode_dev(out, n_b, d_blob, out.t["data_off"], d_out, cap, d_out_off)
    o = d_out_off.cpu().numpy()
    raw = d_out[:te.out_bytes].cpu().numpy().tobytes()
    got = [raw[int(o[b]):int(o[b + 1])] if want_action[b] == 1 else None for b in range(n_b)]
    assert got == pieces


def test_groups_are_partitions(codec):
    """Two partitions in one call: keys are compared within a group only."""
    a = make_partition(5, 60, 9, seed=1)
    b = make_partition(4, 50, 9, seed=2)
    batches = a + b
    group = [0] * len(a) + [1] * len(b)
    got, removed, action, status, _ = codec.compact(b"".join(batches), offsets_of(batches), group)
    want = []
    for part in (a, b):
        st, r, act, pieces = oracle.compact_partition(part)
        want += list(zip(r, act, pieces))
    assert list(zip(removed, action, got)) == want
    # without the groups the two logs are one partition
    st, r, act, pieces = oracle.compact_partition(batches)
    got, removed, action, status, _ = codec.compact(b"".join(batches), offsets_of(batches))
    assert (list(removed), list(action), got) == (list(r), list(act), pieces)


def test_large_partition(codec):
    """A few hundred thousand records, many per key, table probes that collide."""
    batches = make_partition(300, 1500, 20000, seed=9, keyless=0.02)
    check_host(codec, batches)


def test_hot_keys_and_short_batches(codec):
    """Three keys over tens of thousands of records (nearly every record meets an entry whose highest slot
    is already above its own), batches so short that a wave's slots lie in several of them."""
    removed = check_host(codec, make_partition(3000, 37, 3, seed=5, keyless=0.05))
    assert removed.sum() > 50000


def test_full_size_last_of_each_key(codec):
    """At the benchmark's size (8192 batches of 1000 records) the oracle is too slow for a test; the
    domain's own property instead: with record r carrying key r mod K, the records that stay are exactly
    the last K of the log, every batch before them is emptied, the one they begin in is rewritten, and
    compacting what is left removes nothing."""
    import torch
    from tansu_amd import synth
    shape, counts, seed = synth.config("c2", 1.0)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    n_b = len(counts)
    cols, totals, d_blob, d_off = gpu_decode(codec, blob, off, rec_cap=nr, hdr_cap=max(nh, 1))
    assert (totals.n_records, totals.n_failed) == (nr, 0)
    dev = codec.device
    K = 100_003
    count = torch.as_tensor(np.asarray(counts), device=dev).long()
    slot = torch.arange(nr, device=dev)
    key_at = torch.repeat_interleave(cols.t["data_off"][:n_b].long(), count) + cols.field("key_off")[:nr].long()
    assert bool((cols.field("key_len")[:nr] == 16).all())
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    keys = torch.randint(0, 256, (K, 16), dtype=torch.uint8, device=dev, generator=gen)
    for j in range(4):                                     # distinct whatever the draw: the index in four bytes
        keys[:, j] = ((torch.arange(K, device=dev) >> (8 * j)) & 0xFF).to(torch.uint8)
    for j in range(16):
        d_blob[key_at + j] = keys[slot % K, j]
    out, removed, action, t = codec.compact_dev(d_blob, cols, n_b, nr)
    assert t.n_records == K
    base = cols.t["rec_base"][:n_b].long()
    want = torch.clamp(base + count - (nr - K), min=0).minimum(count)
    assert bool((out.t["record_count"][:n_b].long() == want).all())
    assert bool((removed[:n_b].long() == count - want).all())
    want_action = torch.where(want == count, 0, torch.where(want > 0, 1, 2))
    assert bool((action[:n_b].long() == want_action).all())
    # the kept records are the last K, in order
    f, g = cols.field, out.field
    for name in ("offset_delta", "key_off", "key_len", "val_off", "val_len", "timestamp_delta", "rhdr_base"):
        assert bool((g(name)[:K] == f(name)[nr - K:nr]).all()), name
    # what is left, taken as a log again (the operator marks only the rewritten batches OK)
    left = out.t["record_count"][:n_b] > 0
    out.t["status"][:n_b] = torch.where(left, torch.zeros_like(out.t["status"][:n_b]), out.t["status"][:n_b])
    again, removed2, action2, t2 = codec.compact_dev(d_blob, out, n_b, K)
    assert t2.n_records == K and int(removed2.sum()) == 0 and int(action2.sum()) == 0


def test_failed_and_empty(codec):
    batches = make_partition(4, 20, 5, seed=3)
    got, removed, action, status, _ = codec.compact(b"", np.zeros(1, dtype=np.uint64))
    assert got == [] and len(removed) == 0
    bad = list(batches)
    bad[1] = bad[1][:40]                                   # does not decode: takes no part
    got, removed, action, status, _ = codec.compact(b"".join(bad), offsets_of(bad))
    assert status[1] not in (C.OK, hip.NONE) and action[1] == 0
    st, r, act, pieces = oracle.compact_partition([bad[0], bad[2], bad[3]])
    assert [got[0], got[2], got[3]] == pieces
