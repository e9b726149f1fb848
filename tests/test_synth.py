"""The synthetic generator produces batches the CPU restatement accepts, with the
sizes the benchmark configurations call for."""
import numpy as np

import oracle
from tansu_amd import columns as C, synth


def _crc_patch_with_oracle(blob, off):
    crc = np.array([oracle.crc32c(blob[int(off[b]) + 21:int(off[b + 1])].tobytes())
                    for b in range(len(off) - 1)], dtype=np.uint32)
    synth.patch_crc(blob, off, crc)


def test_c1_batch_size_and_decode():
    shape, counts, seed = synth.config("c1", scale=0.001)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    assert int(off[1]) == 11033  # 61 + 64 * 109 + 36 * 111
    _crc_patch_with_oracle(blob, off)
    st, cols, n_rec, n_hdr = oracle.decode_columns(blob, off)
    assert st == C.OK and (cols["status"] == C.OK).all()
    assert (n_rec, n_hdr) == (nr, nh) == (1000, 0)
    assert (cols["crc"] == cols["crc_computed"]).all()
    assert (cols["val_len"] == 100).all() and (cols["key_len"] == -1).all()
    assert list(cols["offset_delta"][:100]) == list(range(100))
    # lengths agree with the fields (canonical input): re-encode is the identity
    st, out, _ = oracle.encode_columns(cols, len(counts), blob, cols["data_off"])
    assert out == blob.tobytes()


def test_c2_shape():
    shape, counts, seed = synth.config("c2", scale=1 / 4096)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    _crc_patch_with_oracle(blob, off)
    st, cols, n_rec, n_hdr = oracle.decode_columns(blob, off)
    assert (cols["status"] == C.OK).all() and (n_rec, n_hdr) == (2000, 4000)
    assert (cols["key_len"] == 16).all() and (cols["val_len"] == 64).all()
    assert (cols["hkey_len"] == 8).all() and (cols["hval_len"] == 16).all()
    assert 141 <= (int(off[1]) - 61) / 1000 <= 143.5


def test_ragged_and_determinism():
    shape, counts, seed = synth.config("c5", scale=1 / 256)
    a = synth.make_batches(shape, counts, seed)
    b = synth.make_batches(shape, counts, seed)
    assert (a[0] == b[0]).all() and 1 <= counts.min() and counts.max() <= 10_000
    blob, off, nr, nh = a
    _crc_patch_with_oracle(blob, off)
    st, cols, n_rec, _ = oracle.decode_columns(blob, off)
    assert (cols["status"] == C.OK).all() and n_rec == nr == int(counts.sum())
    assert list(cols["record_count"]) == list(counts)


def test_mixed_shard_is_a_gpus_share():
    """The mixed workload of one rank: its partitions laid out as one blob whose
    batches the CPU restatement accepts, five shapes in turn, 64 batches each."""
    from tansu_amd import shard
    _, blob, off, nr, nh, nb = shard.make_shard("c4", 10 / 128, 1, 2)
    assert nb == 10 * 64 and len(off) == nb + 1 and int(off[-1]) == len(blob)
    assert (np.diff(off.astype(np.int64)) > 61).all()
    st, cols, got_r, got_h = oracle.decode_columns(blob, off)
    assert st == 0 and (cols["status"] == 0).all() and (got_r, got_h) == (nr, nh)
    # rank 1 of 8 devices holds partitions 1, 9, 17, ...: shapes (p mod 5) 1, 4, 2, 0, 3, ...
    per_batch = cols["record_count"].reshape(10, 64)
    assert (per_batch == per_batch[:, :1]).all()
    assert per_batch[:, 0].tolist()[:5] == [1000, 1, 4, 100, 1]


def test_produce_requests_split_back_into_their_batches():
    """Batches wrapped as ProduceRequest v11 partition data (index, flexible records
    prefix, batches, tagged fields) and split on the host in one call: the starts are
    those the generator placed, and the oracle decodes the buffer with them as it
    decodes the plain blob (what lies between a batch's end and the next batch's
    start, the next field's index and prefix, is not looked at)."""
    from tansu_amd import hip
    shape, counts, seed = synth.config("c5", 0.004)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    buf, field_pos, want_off = synth.make_produce_requests(blob, off, seed + 1)
    assert 1 <= len(field_pos) <= len(counts) and len(buf) > len(blob)
    got_off, first = hip.records_fields_split(buf, field_pos, flexible=True)
    assert (got_off == want_off).all()
    assert first[0] == 0 and first[-1] == len(counts) and (np.diff(first.astype(np.int64)) >= 1).all()
    st, plain, nr1, nh1 = oracle.decode_columns(blob, off)
    st2, framed, nr2, nh2 = oracle.decode_columns(buf, got_off)
    assert st == st2 == 0 and (nr1, nh1) == (nr2, nh2) == (nr, nh)
    for name in ("status", "record_count", "crc_computed", "data_len"):
        assert (plain[name] == framed[name]).all(), name
    assert (framed["data_off"] == got_off[:-1] + 61).all()
    # a field cut short, and a list too short for the batches
    import ctypes
    lib = hip.load()
    n = ctypes.c_uint64(0)
    small = np.zeros(4, dtype=np.uint64)
    assert lib.nisshi_records_fields_split(buf.ctypes.data, len(buf), field_pos.ctypes.data, len(field_pos), 1,
                                           small.ctypes.data, 4, None, ctypes.byref(n)) == 7   # NISSHI_CAPACITY
    full = np.zeros(len(counts) + 2, dtype=np.uint64)
    assert lib.nisshi_records_fields_split(buf.ctypes.data, len(buf) - 100, field_pos.ctypes.data, len(field_pos), 1,
                                           full.ctypes.data, len(full), None, ctypes.byref(n)) == 1  # NISSHI_TRY_GET
