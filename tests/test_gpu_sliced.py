"""The host-buffer decode in slices (copies and kernels of consecutive slices
overlap) gives what one call gives: the oracle's columns.  The slice size comes
from NISSHI_SLICE_BYTES when a context is created; here it is made tiny so that
small inputs are cut into as many slices as the entry point allows."""
import os

import numpy as np
import pytest

import oracle
from tansu_amd import columns as C, hip, synth
from test_gpu_parity import crc_patch, odd_batches
from util import assert_same, find_batches, load_vectors, offsets_of

pytestmark = pytest.mark.gpu

V = load_vectors()


@pytest.fixture(scope="module")
def codec():
    old = os.environ.get("NISSHI_SLICE_BYTES")
    os.environ["NISSHI_SLICE_BYTES"] = "700"
    c = hip.Codec(0)
    if old is None:
        del os.environ["NISSHI_SLICE_BYTES"]
    else:
        os.environ["NISSHI_SLICE_BYTES"] = old
    yield c
    c.close()


def check(codec, blob, off, by_value=True):
    blob = bytes(blob)
    n_b = len(off) - 1
    g, totals = codec.decode(blob, off, by_value)
    st, c, nr, nh = oracle.decode_columns(blob, off, by_value)
    assert st == C.OK
    assert_same(g, c, n_b)
    assert totals.n_failed == int((c["status"] != C.OK).sum())
    assert totals.n_records >= nr and totals.n_headers >= nh
    return g, c, totals


def test_golden_frames_in_slices(codec):
    batches = [bytes(V["batch_71"]["bytes"])]
    for case in V["decode_frames"]:
        frame = bytes(case["frame"])
        batches += [frame[a:b] for a, b in find_batches(frame)]
    batches = batches * 3
    check(codec, b"".join(batches), offsets_of(batches))
    check(codec, b"".join(batches), offsets_of(batches), by_value=False)


def test_odd_and_malformed_batches_in_slices(codec):
    batches = odd_batches()
    check(codec, b"".join(batches), offsets_of(batches))


@pytest.mark.parametrize("name,scale", [("c1", 0.004), ("c2", 0.002), ("c5", 0.01)])
def test_synthetic_in_slices(codec, name, scale):
    shape, counts, seed = synth.config(name, scale)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    crc_patch(codec, blob, off)
    g, c, totals = check(codec, blob, off)
    assert totals.n_records == nr and totals.n_failed == 0
    assert (g["crc"] == g["crc_computed"]).all()


def test_more_slices_than_batches(codec):
    batch = bytes(V["batch_71"]["bytes"])
    check(codec, batch, offsets_of([batch]))          # one batch: one call
    check(codec, batch * 2, offsets_of([batch] * 2))  # two slices of one batch each


def test_pinned_buffers_in_slices(codec):
    shape, counts, seed = synth.config("c2", 8 / 8192)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    crc_patch(codec, blob, off)
    n_b = len(counts)
    p_blob = codec.pinned((len(blob),), np.uint8)
    p_blob[:] = np.frombuffer(bytes(blob), dtype=np.uint8)
    p_off = np.ascontiguousarray(off, dtype=np.uint64)
    arrays = codec.pinned_columns(n_b, nr, nh + 4096 * n_b)
    totals = codec.decode_into(p_blob, p_off, arrays, nr, nh + 4096 * n_b)
    assert (totals.n_records, totals.n_failed) == (nr, 0)
    st, c, _, _ = oracle.decode_columns(blob, off)
    assert_same(C.unpack(arrays), c, n_b)


def test_slices_of_more_batches_than_one_scan_workgroup_takes():
    """Slices of over 32768 batches each: the scan over batches runs a workgroup a
    tile and starts from the totals of the slice before it."""
    old = os.environ.get("NISSHI_SLICE_BYTES")
    os.environ["NISSHI_SLICE_BYTES"] = str(3_600_000)
    codec = hip.Codec(0)
    if old is None:
        del os.environ["NISSHI_SLICE_BYTES"]
    else:
        os.environ["NISSHI_SLICE_BYTES"] = old
    try:
        small = odd_batches()
        for case in V["decode_frames"]:
            frame = bytes(case["frame"])
            small += [frame[a:b] for a, b in find_batches(frame)]
        n_b = 110_000
        batches = [small[(i * 5 + i // len(small)) % len(small)] for i in range(n_b)]
        blob = b"".join(batches)
        assert len(blob) > 2 * 3_600_000 and 3_600_000 / (len(blob) / n_b) > 32768
        check(codec, blob, offsets_of(batches))
    finally:
        codec.close()
