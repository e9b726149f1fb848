"""Parity of the fetch-seam operator (nisshi-storage/src/pg.rs:1875-2052,
lite.rs:2765-2980) with its CPU restatement, through the C ABI.  Bit-exact: the
deflated batches must be the same bytes."""
import numpy as np
import pytest

import oracle
from fetch_cases import CASES, NOW, make_rows
from tansu_amd import columns as C, hip
from util import header_slots

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    c = hip.Codec(0)
    yield c
    c.close()


def fetch_dev(codec, rows, payload, batch_cap=64, built=None):
    """Rows on the GPU -> columns -> encode; returns (status, batches).  `built`: a list
    that receives (batch_length, crc) as inflated::Builder::build stores them."""
    import torch
    n = len(rows["offset"])
    d_rows, keep = codec.fetch_rows_to_device(rows)
    try:
        cols, nb = codec.fetch_batches_dev(d_rows, keep.get("hdr"), n, NOW, batch_cap)
    except hip.NisshiError as e:
        return e.status, []
    pl = np.frombuffer(payload, dtype=np.uint8)
    d_payload = torch.zeros(len(pl) + 16, dtype=torch.uint8, device=codec.device)
    if len(pl):
        d_payload[:len(pl)].copy_(torch.from_numpy(pl.copy()))
    cap = len(pl) + 61 * nb + 64 * n + 32 * len(rows["hdr"]) + 1024
    out = torch.zeros(cap + 16, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(nb + 1, dtype=torch.int64, device=codec.device)
    t = codec.encode_dev(cols, nb, d_payload, cols.t["data_off"], out, cap, out_off)
    off = out_off.cpu().numpy()
    raw = out[:t.out_bytes].cpu().numpy().tobytes()
    if built is not None:
        codec.builder_fields_dev(cols, nb, out, out_off)
        bl = cols.t["batch_length"][:nb].cpu().numpy()
        crc = cols.t["crc"][:nb].cpu().numpy().view(np.uint32)
        built.extend(zip(bl.tolist(), crc.tolist()))
    return C.OK, [raw[int(off[b]):int(off[b + 1])] for b in range(nb)]


@pytest.mark.parametrize("name", sorted(CASES))
def test_cases_device(codec, name):
    kw = dict(CASES[name])
    rows, payload = make_rows(kw.pop("n"), seed=len(name), **kw)
    st, want = oracle.fetch(rows, payload, NOW)
    gst, got = fetch_dev(codec, rows, payload)
    assert gst == st
    assert got == want


@pytest.mark.parametrize("name", sorted(CASES))
def test_cases_host(codec, name):
    kw = dict(CASES[name])
    rows, payload = make_rows(kw.pop("n"), seed=len(name), **kw)
    st, want = oracle.fetch(rows, payload, NOW)
    try:
        got, totals = codec.fetch(rows, payload, NOW)
        gst = C.OK
    except hip.NisshiError as e:
        gst, got = e.status, []
    assert gst == st
    assert got == want


def test_no_rows(codec):
    rows, payload = make_rows(0)
    st, want = oracle.fetch(rows, payload, NOW)
    assert fetch_dev(codec, rows, payload) == (C.OK, want)
    got, _ = codec.fetch(rows, payload, NOW)
    assert got == want


def test_capacity(codec):
    rows, payload = make_rows(30, seed=1, change_every=3)
    st, _ = fetch_dev(codec, rows, payload, batch_cap=4)
    assert st == C.CAPACITY
    st, want = oracle.fetch(rows, payload, NOW, batch_cap=10)
    assert fetch_dev(codec, rows, payload, batch_cap=10) == (st, want)


@pytest.mark.parametrize("kw", [dict(n=20000, change_every=997, gap_rate=0.2),
                                dict(n=30000, change_every=4100, open_rate=0.0002, gap_rate=0.5),
                                dict(n=9000, change_every=2, headers=3)])
def test_many_rows(codec, kw):
    # several blocks of the batch scan, batches that straddle them
    kw = dict(kw)
    rows, payload = make_rows(kw.pop("n"), seed=5, **kw)
    st, want = oracle.fetch(rows, payload, NOW, batch_cap=8192)
    gst, got = fetch_dev(codec, rows, payload, batch_cap=8192)
    assert gst == st == C.OK
    assert got == want


def test_fetch_leaves_the_batches_sizes_for_the_encoder(codec):
    """The fetch seam works out every record's length: the sum a batch is left in `enc_len` (bytes + 1) so that the
    encode does not size the records again -- for batches of fewer than 2048 records; larger ones, which are
    written in pieces, are sized by the encoder."""
    rows, payload = make_rows(20000, seed=9, change_every=1700, gap_rate=0.1)
    big, payload_big = make_rows(9000, seed=10, change_every=4000)
    for r, pl in ((rows, payload), (big, payload_big)):
        st, want = oracle.fetch(r, pl, NOW, batch_cap=8192)
        d_rows, keep = codec.fetch_rows_to_device(r)
        cols, nb = codec.fetch_batches_dev(d_rows, keep.get("hdr"), len(r["offset"]), NOW, 8192)
        assert st == C.OK and nb == len(want)
        enc_len = cols.t["enc_len"][:nb].cpu().numpy().view(np.uint32)
        count = cols.t["record_count"][:nb].cpu().numpy()
        expect = np.array([len(w) - 61 + 1 for w in want], dtype=np.uint32)
        expect[count >= 2048] = 0
        assert (enc_len == expect).all() and (enc_len != 0).any()
    assert (expect == 0).any()


def test_produce_then_fetch_round_trip(codec):
    """Rows written by the produce seam and read back by the fetch seam carry the
    same records: the batch's records at offsets high .. high + n - 1."""
    from test_gpu_storage import header_batches
    batch = header_batches()[0]
    blob = batch
    off = np.array([0, len(batch)], dtype=np.uint64)
    high = np.array([4200], dtype=np.int64)
    g, prow, totals = codec.produce_rows(blob, off, high, True)
    n = int(g["record_count"][0])
    base = int(g["data_off"][0])
    hdr = np.stack([g["hkey_off"] + base, g["hkey_len"].view(np.uint32), g["hval_off"] + base,
                    g["hval_len"].view(np.uint32)], axis=1).astype(np.uint32)
    counts = g["rhdr_count"][:n].astype(np.int64)
    hb = np.concatenate([[0], np.cumsum(counts)])
    slots = header_slots(g["rhdr_base"][:n], counts)   # a record's headers are contiguous from its slot
    rows = dict(offset=prow["offset"][:n], attributes=np.full(n, g["attributes"][0], dtype=np.int16),
                timestamp=prow["timestamp"][:n], producer_id=np.full(n, g["producer_id"][0]),
                producer_epoch=np.full(n, g["producer_epoch"][0], dtype=np.int16), completed=np.ones(n),
                key_off=g["key_off"][:n] + base, key_len=g["key_len"][:n],
                val_off=g["val_off"][:n] + base, val_len=g["val_len"][:n],
                hdr_base=hb, hdr=hdr[slots])
    got, _ = codec.fetch(rows, blob, NOW)
    assert len(got) == 1
    st, cols, nr, nh = oracle.decode_columns(got[0], np.array([0, len(got[0])], dtype=np.uint64), True)
    assert st == C.OK and int(cols["base_offset"][0]) == 4200
    back = C.records_of_batch(cols, 0, got[0])
    st, c0, _, _ = oracle.decode_columns(batch, off, True)
    orig = C.records_of_batch(c0, 0, batch)
    assert len(back) == len(orig) == n
    first_ts = orig[0].timestamp_delta
    for a, b in zip(orig, back):
        assert (a.key, a.value, a.headers) == (b.key, b.value, b.headers)
        assert a.timestamp_delta - first_ts == b.timestamp_delta


def test_full_size_produce_then_fetch(codec):
    """At the benchmark's size (8192 batches of 1000 small records, two headers each) the oracle is too
    slow for a test; the round trip instead: the rows the produce seam writes for the decoded batches,
    read back by the fetch seam and deflated, are the input again byte for byte but for the 61-byte batch
    headers (the records' lengths and deltas are canonical in this workload, each batch's rows have a
    producer id of their own, the deltas count from the batch's first row as they did), and the rebuilt
    headers carry the offsets the rows were given."""
    import torch
    from tansu_amd import synth
    from test_gpu_parity import gpu_decode
    dev = codec.device
    shape, counts, seed = synth.config("c2", 1.0)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    n_b = len(counts)
    cols, totals, d_blob, d_off = gpu_decode(codec, blob, off, rec_cap=nr, hdr_cap=max(nh, 1))
    assert (totals.n_records, totals.n_headers, totals.n_failed) == (nr, nh, 0)
    per = int(counts[0])
    high = torch.arange(n_b, device=dev, dtype=torch.int64) * per + 7_000_000
    rows = codec.produce_rows_dev(cols, n_b, nr, high, True)
    assert bool((rows.t["offset"][:nr] == torch.arange(nr, device=dev) + 7_000_000).all())
    # the rows a storage engine would hand back: spans as offsets into the blob, headers in record order
    rkey = cols.t["rkey"][:nr].view(torch.int32).reshape(nr, 4)
    rval = cols.t["rval"][:nr].view(torch.int32).reshape(nr, 4)
    batch = rows.t["batch"][:nr].long()
    base = cols.t["data_off"][:n_b][batch]
    hc = rval[:, 3].long()
    hb = torch.zeros(nr + 1, dtype=torch.int64, device=dev)
    hb[1:] = torch.cumsum(hc, 0)
    assert int(hb[-1]) == nh
    within = torch.arange(nh, device=dev) - torch.repeat_interleave(hb[:-1], hc)
    src = torch.repeat_interleave(rval[:, 2].long() & 0xffffffff, hc) + within
    hdr = cols.t["hdr"].view(torch.int32).reshape(-1, 4)[src].clone()
    hbase = torch.repeat_interleave(base, hc).int()
    hdr[:, 0] += hbase
    hdr[:, 2] += hbase
    del within, src, hbase
    keep = dict(offset=rows.t["offset"][:nr].clone(), attributes=torch.zeros(nr, dtype=torch.int16, device=dev),
                timestamp=rows.t["timestamp"][:nr].clone(), producer_id=batch.clone(),
                producer_epoch=torch.full((nr,), -1, dtype=torch.int16, device=dev),
                completed=torch.ones(nr, dtype=torch.uint8, device=dev),
                key_off=(rkey[:, 2].long() + base).int(), key_len=rkey[:, 3].contiguous(),
                val_off=(rval[:, 0].long() + base).int(), val_len=rval[:, 1].contiguous(),
                hdr_base=hb.int(), hdr=hdr.contiguous())
    c = hip.CFetchRows()
    for name, _, _ in hip.FETCH_ARRAYS:
        setattr(c, name, keep[name].data_ptr())
    fcols, nb = codec.fetch_batches_dev(c, keep["hdr"], nr, NOW, n_b + 1)
    assert nb == n_b
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=dev)
    out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=dev)
    te = codec.encode_dev(fcols, nb, d_blob, fcols.t["data_off"], out, len(blob), out_off)
    assert te.out_bytes == len(blob)
    assert bool((out_off.cpu() == torch.from_numpy(off.astype(np.int64))).all())
    body = torch.ones(len(blob), dtype=torch.bool, device=dev)
    starts = d_off[:n_b].long()
    for k in range(61):
        body[starts + k] = False
    assert bool((out[:len(blob)][body] == d_blob[:len(blob)][body]).all())
    del body
    back = hip.DeviceColumns(n_b, nr, max(nh, 1), dev)
    t2 = codec.decode_dev(out, len(blob), out_off, n_b, back)
    assert (t2.n_records, t2.n_headers, t2.n_failed) == (nr, nh, 0)
    bt = back.t
    assert bool((bt["crc"][:n_b] == bt["crc_computed"][:n_b]).all())
    assert bool((bt["base_offset"][:n_b] == high).all())
    assert bool((bt["record_count"][:n_b] == per).all()) and bool((bt["last_offset_delta"][:n_b] == per - 1).all())
    assert bool((bt["producer_id"][:n_b] == torch.arange(n_b, device=dev)).all())
    first = rows.t["timestamp"][:nr].view(n_b, per)
    assert bool((bt["base_timestamp"][:n_b] == first[:, 0]).all())
    assert bool((bt["max_timestamp"][:n_b] == NOW).all())    # the builder's default, as the reference leaves it


def builder_fields_of(batch: bytes):
    """inflated::Builder::build's batch_length and crc for a deflated batch, by the
    CPU restatement: inflate it, into_builder(), build()."""
    st, cols, _, _ = oracle.decode_columns(batch, np.array([0, len(batch)], dtype=np.uint64), True)
    assert st == C.OK
    st, bl, crc = oracle.builder_columns(cols, 1, batch, cols["data_off"])
    assert st == C.OK
    return int(bl[0]), int(crc[0])


@pytest.mark.parametrize("kw", [dict(n=1), dict(n=300, change_every=40, headers=2),
                                dict(n=5000, change_every=700, gap_rate=0.3)])
def test_builder_fields(codec, kw):
    """nisshi_hip_builder_fields_dev after fetch + encode: the inflated::Batch that the
    reference's Builder::build returns has these batch_length and crc.  Values of
    55-130 bytes put many records' body sizes in 64..127, where Sequence::size_in_bytes
    counts the length prefix a byte short (codec.rs:482-491)."""
    kw = dict(kw)
    rows, payload = make_rows(kw.pop("n"), seed=9, **kw)
    built = []
    st, got = fetch_dev(codec, rows, payload, batch_cap=256, built=built)
    assert st == C.OK and len(built) == len(got)
    short = 0
    for (bl, crc), batch in zip(built, got):
        assert (bl, crc) == builder_fields_of(batch)
        wire_bl = int.from_bytes(batch[8:12], "big", signed=True)
        assert crc == int.from_bytes(batch[17:21], "big") and bl <= wire_bl
        short += wire_bl - bl
    assert kw == {} or short > 0


def test_builder_fields_golden(codec):
    """The reference's own case (inflated.rs:533-554): one record with value `def`
    builds to batch_length 59 and crc 1126819645."""
    import torch
    from util import load_vectors
    raw = bytes(load_vectors()["batch_71"]["bytes"])
    off = np.array([0, len(raw)], dtype=np.uint64)
    d_blob, d_off = codec.to_device(np.frombuffer(raw, dtype=np.uint8), off)
    cols = hip.DeviceColumns(1, 1, 1, codec.device)
    codec.decode_dev(d_blob, len(raw), d_off, 1, cols)
    out = torch.zeros(len(raw) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(2, dtype=torch.int64, device=codec.device)
    codec.encode_dev(cols, 1, d_blob, cols.t["data_off"], out, len(raw), out_off)
    cols.t["batch_length"].zero_()
    cols.t["crc"].zero_()
    codec.builder_fields_dev(cols, 1, out, out_off)
    assert int(cols.t["batch_length"][0]) == 59
    assert int(cols.t["crc"][0].cpu().numpy().view(np.uint32)) == 1126819645
