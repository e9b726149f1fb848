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
