"""nisshi_hip_pool, the scheduler of one host frame over several devices, without
a GPU: a pool of fake devices whose decoder is the CPU checker (oracle/).  What
is under test is the pool's own work: batch b goes to device partition[b] mod G,
every device gets its batches in input order as one blob, and the columns that
come back are, batch for batch, what one decode of the whole frame gives, with
record and header slots renumbered into the caller's arrays."""
import ctypes

import numpy as np
import pytest

import oracle
from tansu_amd import build, columns as C, hip, synth
from test_gpu_parity import odd_batches
from util import assert_same, offsets_of


@pytest.fixture(scope="module")
def lib():
    build.build()
    return hip.load()


def frame():
    """Batches of several kinds and sizes, malformed ones among them, and a partition id a batch."""
    parts = synth.mixed_partitions(10, 3)
    batches = []
    partition = []
    for p, (blob, off, _, _) in enumerate(parts):
        for i in range(len(off) - 1):
            batches.append(bytes(blob[int(off[i]):int(off[i + 1])]))
            partition.append(p)
    for i, b in enumerate(odd_batches()):
        batches.insert(3 * i, b)
        partition.insert(3 * i, 100 + i % 7)
    return batches, np.array(partition, dtype=np.uint32)


def oracle_decoder(seen):
    """A decoder for nisshi_hip_pool_create_custom: the CPU checker, noting what each device was given."""
    def decode(ctx, blob, blob_len, batch_off, n_batches, by_value, cols, rec_cap, hdr_cap, totals):
        off = np.ctypeslib.as_array(ctypes.cast(batch_off, ctypes.POINTER(ctypes.c_uint64)), shape=(n_batches + 1,)).copy()
        data = ctypes.string_at(blob, blob_len)
        seen.append((ctx - 1, data, off))
        st, got, nr, nh = oracle.decode_columns(data, off, bool(by_value), rec_cap=max(rec_cap, 1), hdr_cap=max(hdr_cap, 1))
        # the checker's columns (an array a field) into the caller's (the library's groups of fields)
        packed = C.pack(got)
        dst = ctypes.cast(cols, ctypes.POINTER(C.CColumns)).contents
        for name, dt, words, level in C.DEVICE_ARRAYS:
            n = (n_batches, nr, nh)[level]
            if n:
                src = np.ascontiguousarray(packed[name][:n])
                ctypes.memmove(getattr(dst, name), src.ctypes.data, src.nbytes)
        t = ctypes.cast(totals, ctypes.POINTER(hip.Totals)).contents
        t.n_records, t.n_headers = nr, nh
        t.n_failed, t.n_exact, t.out_bytes = int((got["status"][:n_batches] != 0).sum()), 0, 0
        return st
    return decode


@pytest.mark.parametrize("g", [1, 2, 3, 8])
def test_frame_over_fake_devices_equals_one_decode(lib, g):
    batches, partition = frame()
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8)
    off = offsets_of(batches)
    n_b = len(batches)
    st, want, nr, nh = oracle.decode_columns(blob, off)
    assert st == C.OK
    seen = []
    pool = hip.Pool(g, decoder=oracle_decoder(seen))
    try:
        assert [pool.device_of(p) for p in (0, 1, g, g + 1, 1023)] == [0, 1 % g, 0, (g + 1) % g, 1023 % g]
        got, totals, per_device = pool.decode(blob, off, partition, nr, max(nh, 1))
    finally:
        pool.close()
    # every device got its batches, in input order, as one blob
    assert sorted(d for d, _, _ in seen) == [d for d in range(g) if (partition % g == d).any()]
    for d, data, doff in seen:
        mine = [batches[b] for b in range(n_b) if partition[b] % g == d]
        assert data == b"".join(mine)
        assert list(doff) == list(offsets_of(mine))
    # the columns are those of one decode of the whole frame, batch for batch
    assert (totals.n_records, totals.n_headers) == (nr, nh)
    assert totals.n_failed == int((want["status"] != C.OK).sum()) == sum(t.n_failed for t in per_device)
    assert_same(got, want, n_b)
    assert (got["data_off"][:n_b] == want["data_off"][:n_b]).all()
    # a device's records are one run of the record arrays, devices in ascending order
    ok = want["status"][:n_b] == C.OK
    dev = (partition % g)[ok]
    base = got["rec_base"][:n_b][ok].astype(np.int64)
    order = np.lexsort((np.arange(len(dev)), dev))
    assert (np.diff(base[order]) >= 0).all()


def test_capacity_and_arguments(lib):
    batches, partition = frame()
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8)
    off = offsets_of(batches)
    st, want, nr, nh = oracle.decode_columns(blob, off)
    pool = hip.Pool(2, decoder=oracle_decoder([]))
    try:
        with pytest.raises(hip.NisshiError) as e:
            pool.decode(blob, off, partition, nr, nh // 2)
        assert e.value.status in (C.CAPACITY,)
        got, totals, _ = pool.decode(blob[:0], off[:1], partition[:0], 1, 1)     # an empty frame
        assert (totals.n_records, totals.n_headers, totals.n_failed) == (0, 0, 0)
    finally:
        pool.close()


def test_real_pool_needs_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hip.NisshiError):
        hip.Pool([0])
