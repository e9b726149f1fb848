"""A receive buffer of produce requests on the device: the batches of C5 wrapped as
ProduceRequest v11 partition data (index, flexible records prefix, 1-4 batches,
tagged fields), found again by the host split (the reference's Decoder:
nisshi-sans-io/src/de.rs:701-720, 862-876, 1056-1093) and decoded where they
lie -- what is between a batch's end and the next batch's start never reaches a
record.  Against the CPU checker's decode of the same batches."""
import ctypes
import struct

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


def requests(scale=0.02, seed_add=1):
    shape, counts, seed = synth.config("c5", scale)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    buf, field_pos, want_off = synth.make_produce_requests(blob, off, seed + seed_add)
    return blob, off, buf, field_pos, want_off, nr, nh


def test_request_buffer_decodes_as_its_batches(codec):
    blob, off, buf, field_pos, want_off, nr, nh = requests()
    got_off, first = hip.records_fields_split(buf, field_pos, flexible=True)
    n_b = len(got_off) - 1
    assert (got_off == want_off).all() and n_b == len(off) - 1
    # every field's last batch is followed by bytes that are not a batch (the tagged-field
    # section, the next partition's index and prefix): its extent in got_off is longer than
    # 12 + batch_length, and batch_length alone says where its record data ends
    lens = np.diff(got_off.astype(np.int64))
    bl = np.array([struct.unpack_from(">i", buf, int(o) + 8)[0] for o in got_off[:-1]], dtype=np.int64)
    last_of_field = first[1:-1].astype(np.int64) - 1     # (the last field's end is the last entry of got_off)
    assert (lens[last_of_field] > bl[last_of_field] + 12).all() and len(last_of_field) > 10
    inner = np.setdiff1d(np.arange(n_b), last_of_field)
    assert (lens[inner] == bl[inner] + 12).all()

    st, want, wr, wh = oracle.decode_columns(buf, got_off)
    assert st == C.OK and (wr, wh) == (nr, nh) and (want["status"][:n_b] == C.OK).all()
    d_buf, d_off = codec.to_device(buf, got_off)
    cols = hip.DeviceColumns(n_b, nr, max(nh, 1), codec.device)
    totals = codec.decode_dev(d_buf, len(buf), d_off, n_b, cols)
    assert (totals.n_records, totals.n_headers, totals.n_failed) == (nr, nh, 0)
    got = cols.to_host(nr, nh)
    assert_same(got, want, n_b)
    assert (got["data_off"][:n_b] == got_off[:-1] + 61).all()
    assert (got["data_len"][:n_b] == bl - 49).all()
    # the same records as the plain blob of the same batches gives
    st, plain, _, _ = oracle.decode_columns(blob, off)
    for name in ("record_count", "crc_computed", "data_len", "last_offset_delta"):
        assert (got[name][:n_b] == plain[name][:n_b]).all(), name
    # and back to the wire: the batches, without what lay between them
    import torch
    out = torch.zeros(len(blob) + 64, dtype=torch.uint8, device=codec.device)
    out_off = torch.zeros(n_b + 1, dtype=torch.int64, device=codec.device)
    # (crc fields as the generator left them: the encode recomputes crc, so compare all but those four bytes)
    te = codec.encode_dev(cols, n_b, d_buf, cols.t["data_off"], out, len(blob), out_off)
    assert te.out_bytes == len(blob)
    back = out[:len(blob)].cpu().numpy()
    mask = np.ones(len(blob), dtype=bool)
    for k in range(4):
        mask[off[:-1].astype(np.int64) + 17 + k] = False
    assert (back[mask] == blob[mask]).all()
    assert (got["crc_computed"][:n_b] ==
            np.array([struct.unpack_from(">I", back, int(o) + 17)[0] for o in off[:-1]], dtype=np.uint32)).all()


def test_host_split_refuses_a_batch_that_runs_past_its_field(codec):
    """A field's last batch whose batch_length reaches into what follows the field: the
    reference's split fails with Overflow when batch_length exceeds what is left of the
    records blob (de.rs:1072-1080); such offsets never reach the device."""
    blob, off, buf, field_pos, want_off, nr, nh = requests(0.01, 2)
    got_off, first = hip.records_fields_split(buf, field_pos, flexible=True)
    b = int(first[1]) - 1                       # the last batch of the first field
    bad = buf.copy()
    bl = struct.unpack_from(">i", bad, int(got_off[b]) + 8)[0]
    struct.pack_into(">i", bad, int(got_off[b]) + 8, bl + 1)   # one byte into the tagged-field section
    lib = hip.load()
    n = ctypes.c_uint64(0)
    full = np.zeros(len(got_off) + 2, dtype=np.uint64)
    st = lib.nisshi_records_fields_split(bad.ctypes.data, len(bad), field_pos.ctypes.data, len(field_pos), 1,
                                         full.ctypes.data, len(full), None, ctypes.byref(n))
    assert st == C.OVERFLOW
    # one byte short instead: the rest of the field is not a whole batch header either
    struct.pack_into(">i", bad, int(got_off[b]) + 8, bl - 1)
    st = lib.nisshi_records_fields_split(bad.ctypes.data, len(bad), field_pos.ctypes.data, len(field_pos), 1,
                                         full.ctypes.data, len(full), None, ctypes.byref(n))
    assert st != C.OK
    # the untouched buffer decodes; the device is never given offsets the split refused, and
    # the host-buffer entry point refuses offsets that are not ascending or leave the buffer
    arrays = C.alloc_device_layout(len(got_off) - 1, nr, max(nh, 1))
    t = codec.decode_into(buf, got_off, arrays, nr, max(nh, 1))
    assert (t.n_records, t.n_failed) == (nr, 0)
    wrong = got_off.copy()
    wrong[3] = wrong[2] - 1
    with pytest.raises(hip.NisshiError) as e:
        codec.decode_into(buf, wrong, arrays, nr, max(nh, 1))
    assert e.value.status == 9    # NISSHI_INVALID_ARGUMENT
    wrong = got_off.copy()
    wrong[-1] = len(buf) + 1
    with pytest.raises(hip.NisshiError) as e:
        codec.decode_into(buf, wrong, arrays, nr, max(nh, 1))
    assert e.value.status == 9
