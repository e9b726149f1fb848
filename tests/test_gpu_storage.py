"""Parity of the column operators for the callers of the codec in nisshi-storage
(the SQL engines' produce seam, pg.rs:797-943 and lite.rs:703-815, and combine,
batch.rs:661-709) with their CPU restatement, through the C ABI.  Bit-exact."""
import struct

import numpy as np
import pytest

import oracle
from tansu_amd import columns as C, hip, synth
from test_gpu_parity import crc_patch, gpu_decode, odd_batches
from test_storage_ops import REF_BATCHES, into_batches
from util import build_batch, find_batches, load_vectors, offsets_of

pytestmark = pytest.mark.gpu

V = load_vectors()


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def to_dev(codec, arr, dtype):
    import torch
    a = np.ascontiguousarray(arr, dtype=dtype)
    signed = a.view(np.dtype(f"i{a.dtype.itemsize}"))
    return torch.from_numpy(signed.copy()).to(codec.device) if len(a) else \
        torch.zeros(1, dtype=torch.int64, device=codec.device)


def golden_batches():
    out = [bytes(V["batch_71"]["bytes"])]
    for case in V["decode_frames"]:
        frame = bytes(case["frame"])
        out += [frame[a:b] for a, b in find_batches(frame)]
    return out


def header_batches():
    recs = [C.Record(0, 0, 5, 0, b"k0", b"v0", [C.Header(b"a", b"1"), C.Header(b"b", None)]),
            C.Record(0, 0, 6, 1, None, None, []),
            C.Record(0, 0, -3, 2, b"", b"x" * 300, [C.Header(None, b"z")])]
    neg = [C.Record(0, 0, 0, 0, None, b"a"), C.Record(0, 0, -11, 1, None, b"b")]
    return [build_batch(dict(base_timestamp=1_700_000_000_000, last_offset_delta=2), recs),
            build_batch(dict(base_timestamp=10, last_offset_delta=1), neg),
            build_batch(dict(base_timestamp=-5, last_offset_delta=0), neg[:1]),
            build_batch(dict(base_timestamp=(1 << 63) - 1, last_offset_delta=1),
                        [C.Record(0, 0, 1, 0, None, b"wraps")])]


# ---------------------------------------------------------------- produce rows


def check_rows(codec, batches, check_time, via_host=False):
    blob = b"".join(batches)
    off = offsets_of(batches)
    n_b = len(batches)
    rng = np.random.default_rng(11)
    high = rng.integers(0, 1 << 40, size=n_b, dtype=np.int64)
    high[::3] = 0                                           # high = None (unwrap_or_default)
    if via_host:
        g, rows, totals = codec.produce_rows(blob, off, high, check_time)
    else:
        cols, totals, d_blob, d_off = gpu_decode(codec, blob, off, hdr_cap=len(blob) // 2 + 4096 * n_b + 1)
        d_rows = codec.produce_rows_dev(cols, n_b, totals.n_records, to_dev(codec, high, np.int64), check_time)
        g = cols.to_host()
        rows = d_rows.to_host()
    checked = 0
    for b, batch in enumerate(batches):
        st, attributes, want, hwant = oracle.produce_rows(batch, int(high[b]), check_time)
        assert int(rows["status"][b]) == st, (b, st)
        if st != C.OK:
            continue
        assert int(g["attributes"][b]) == attributes
        n = len(want["offset"])
        assert int(g["record_count"][b]) == n
        r0 = int(g["rec_base"][b])
        sl = slice(r0, r0 + n)
        assert (rows["offset"][sl] == want["offset"]).all()
        assert (rows["timestamp"][sl] == want["timestamp"]).all()
        assert (rows["batch"][sl] == b).all()
        # spans: the decode's, relative to the batch's record data (61 bytes in)
        assert (g["key_len"][sl] == want["key_len"]).all() and (g["val_len"][sl] == want["val_len"]).all()
        assert (g["key_off"][sl].astype(np.int64) + 61 == want["key_pos"].astype(np.int64)).all()
        assert (g["val_off"][sl].astype(np.int64) + 61 == want["val_pos"].astype(np.int64)).all()
        h = 0
        for r in range(r0, r0 + n):
            for slot in range(int(g["rhdr_base"][r]), int(g["rhdr_base"][r]) + int(g["rhdr_count"][r])):
                assert int(rows["hdr_offset"][slot]) == int(hwant["offset"][h])
                assert int(g["hkey_off"][slot]) + 61 == int(hwant["key_pos"][h])
                assert int(g["hkey_len"][slot]) == int(hwant["key_len"][h])
                assert int(g["hval_off"][slot]) + 61 == int(hwant["val_pos"][h])
                assert int(g["hval_len"][slot]) == int(hwant["val_len"][h])
                h += 1
        assert h == len(hwant["offset"])
        checked += n
    return checked, rows


@pytest.mark.parametrize("check_time", [True, False])
def test_produce_rows_golden_and_odd(codec, check_time):
    batches = golden_batches() + header_batches() + odd_batches()
    checked, rows = check_rows(codec, batches, check_time)
    assert checked > 100
    neg = len(golden_batches()) + 1
    assert int(rows["status"][neg]) == (C.TRY_FROM_INT if check_time else C.OK)
    checked_host, _ = check_rows(codec, batches, check_time, via_host=True)
    assert checked_host == checked


@pytest.mark.parametrize("name,scale", [("c1", 0.02), ("c2", 0.002), ("c5", 0.02)])
def test_produce_rows_synthetic(codec, name, scale):
    shape, counts, seed = synth.config(name, scale)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    crc_patch(codec, blob, off)
    batches = [blob[int(off[b]):int(off[b + 1])].tobytes() for b in range(len(counts))]
    checked, _ = check_rows(codec, batches, True)
    assert checked == nr


# ---------------------------------------------------------------- combine


def gpu_combine(codec, batches, group_off):
    """decode -> combine_dev -> encode_dev; returns (bytes or None per group, status)."""
    import torch
    blob = b"".join(batches)
    off = offsets_of(batches)
    n_b, n_g = len(batches), len(group_off) - 1
    cols, totals, d_blob, d_off = gpu_decode(codec, blob, off, hdr_cap=len(blob) // 2 + 4096 * n_b + 1)
    d_groups = to_dev(codec, group_off, np.uint32)
    sink = codec.combine_dev(cols, n_b, totals.n_records, d_groups, n_g)
    # the rebased deltas may need longer varints: up to 13 bytes more a record
    out = torch.zeros(len(blob) + 13 * totals.n_records + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n_g + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(sink, n_g, d_blob, sink.t["data_off"], out, len(out), out_off)
    o = out_off.cpu().numpy()
    data = out[:t.out_bytes].cpu().numpy().tobytes()
    status = sink.t["status"].cpu().numpy()[:n_g]
    return [data[int(o[g]):int(o[g + 1])] if status[g] == C.OK else None for g in range(n_g)], status


def check_combine(codec, batches, group_off):
    got, status = gpu_combine(codec, batches, group_off)
    host, host_status, _ = codec.combine(b"".join(batches), offsets_of(batches), group_off)
    for g in range(len(group_off) - 1):
        members = batches[group_off[g]:group_off[g + 1]]
        st, want = oracle.combine(members)
        if not members:
            assert status[g] == hip.NONE and host_status[g] == hip.NONE and want is None
            continue
        assert status[g] == st and host_status[g] == st, (g, status[g], st)
        if st == C.OK:
            assert got[g] == want, g
            assert host[g] == want, g
    return got, status


def test_combine_reference_case(codec):
    # batch.rs:1244-1349 through the GPU: one group of the six batches
    batches = into_batches(0, 54345, 32123, 0, REF_BATCHES, [1_000, 1_250, 900, 1_000, 70_000, 5])
    got, status = check_combine(codec, batches, [0, 6])
    g, totals = codec.decode(got[0], offsets_of([got[0]]))
    records = C.records_of_batch(g, 0, got[0])
    want = [v for values in REF_BATCHES for v in values]
    assert [r.value for r in records] == want
    assert [r.offset_delta for r in records] == list(range(12))
    assert all(r.key is None for r in records)
    assert (int(g["producer_id"][0]), int(g["producer_epoch"][0]), int(g["base_sequence"][0])) == (54345, 32123, 0)
    # batch.rs:1198-1202: nothing in, nothing out
    got, status = check_combine(codec, batches, [0, 0, 6, 6])
    assert list(status) == [hip.NONE, C.OK, hip.NONE] and got[0] is None and got[2] is None


def test_combine_groups_of_many_shapes(codec):
    rng = np.random.default_rng(5)
    batches = []
    for blob, off, nr, nh in synth.mixed_partitions(6, 40):
        crc_patch(codec, blob, off)
        batches += [blob[int(off[b]):int(off[b + 1])].tobytes() for b in range(len(off) - 1)]
    batches += header_batches() + golden_batches()
    # vary what combine reads: last_offset_delta, timestamps
    for i, b in enumerate(batches):
        b = bytearray(b)
        b[23:27] = struct.pack(">i", int(rng.integers(0, 50)))
        b[27:35] = struct.pack(">q", int(rng.integers(0, 1 << 41)))
        b[35:43] = struct.pack(">q", int(rng.integers(0, 1 << 41)))
        batches[i] = bytes(b)
    n = len(batches)
    cuts = sorted({c for c in rng.integers(0, n, size=30).tolist() if not 70 < c < 200} | {0, n, 70, 200})
    group_off = [0]
    for c in cuts[1:]:
        group_off.append(c)
        if rng.integers(0, 4) == 0:
            group_off.append(c)                                    # an empty group
    assert max(np.diff(group_off)) > 64                            # more members than one wave pass
    check_combine(codec, batches, group_off)


def test_combine_with_members_that_fail(codec):
    good = into_batches(0x30, 7, 1, 100, REF_BATCHES)
    odd = odd_batches()
    batches = good[:3] + [odd[8]] + good[3:] + [odd[6], good[0], odd[24], good[1]] + good
    group_off = [0, 7, 9, 11, len(batches)]
    got, status = check_combine(codec, batches, group_off)
    assert status[0] != C.OK and status[1] != C.OK and status[2] == C.COMPRESSED and status[3] == C.OK
    # a failed group leaves its members as they were: combining them alone still works
    check_combine(codec, batches, [0, 3, 3, 4, 7, 8, 9, 10, 11, len(batches)])


def test_combine_full_size_properties(codec):
    """8192 batches of 1000 records in groups of eight: the sinks decode again to
    1024 batches whose offset deltas count up across the members."""
    import torch
    shape, counts, seed = synth.config("c2", 0.25)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    crc_patch(codec, blob, off)
    n_b = len(counts)
    r = int(counts[0])
    assert n_b % 8 == 0
    cols, totals, d_blob, d_off = gpu_decode(codec, blob, off, rec_cap=nr, hdr_cap=max(nh, 1))
    n_g = n_b // 8
    d_groups = to_dev(codec, np.arange(0, n_b + 1, 8), np.uint32)
    sink = codec.combine_dev(cols, n_b, totals.n_records, d_groups, n_g)
    assert bool((sink.t["status"] == 0).all())
    assert bool((sink.t["record_count"] == 8 * r).all())
    lod = cols.t["last_offset_delta"].view(n_g, 8).sum(dim=1) + 7
    assert bool((sink.t["last_offset_delta"] == lod).all())
    out = torch.zeros(len(blob) + 2 * nr + 64, dtype=torch.uint8, device=codec.device)   # longer offset deltas
    out_off = torch.zeros(n_g + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(sink, n_g, d_blob, sink.t["data_off"], out, len(out) - 16, out_off)
    o = out_off.cpu().numpy().view(np.uint64)
    # one group against the CPU restatement, byte for byte
    g = n_g // 2
    members = [blob[int(off[b]):int(off[b + 1])].tobytes() for b in range(8 * g, 8 * g + 8)]
    st, want = oracle.combine(members)
    assert st == C.OK and out[int(o[g]):int(o[g + 1])].cpu().numpy().tobytes() == want
    # and all of them through a second decode
    # (the records whose offset delta grew keep their stored length, so these batches
    # end on the sequential walk, which starts over with fresh header slots)
    cols2 = hip.DeviceColumns(n_g, nr, 2 * nh + 4096 * n_g, codec.device)
    t2 = codec.decode_dev(out, t.out_bytes, out_off, n_g, cols2)
    assert (t2.n_records, t2.n_failed) == (nr, 0), cols2.t["status"].unique()
    assert bool((cols2.t["crc"] == cols2.t["crc_computed"]).all())
    od = cols2.field("offset_delta")[:nr].view(n_g, 8, r)
    step = (cols.t["last_offset_delta"].view(n_g, 8) + 1).cumsum(dim=1) - (cols.t["last_offset_delta"].view(n_g, 8) + 1)
    want_od = torch.arange(r, device=od.device, dtype=od.dtype).view(1, 1, r) + step.view(n_g, 8, 1).to(od.dtype)
    assert bool((od == want_od).all())
    # produce rows over the sinks: offsets count up from each group's high watermark
    high = torch.arange(n_g, device=codec.device, dtype=torch.int64) * 1_000_000
    rows = codec.produce_rows_dev(cols2, n_g, nr, high, True)
    want_off = (high.view(n_g, 1) + torch.arange(8 * r, device=codec.device).view(1, -1)).view(-1)
    assert bool((rows.t["offset"][:nr] == want_off).all())
    ts = cols2.t["base_timestamp"].view(n_g, 1) + cols2.field("timestamp_delta")[:nr].view(n_g, 8 * r)
    assert bool((rows.t["timestamp"][:nr] == ts.view(-1)).all())
    assert bool((rows.t["status"] == 0).all())
