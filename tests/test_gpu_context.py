"""A context serves one call at a time, whatever streams the calls are given: a
call that only enqueues work is followed on the device by the next call, also on
another stream, and by a call that has to grow the workspace.  Column pointers
the kernels cannot use are refused."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from tansu_amd import columns as C, hip, synth
from util import assert_same

pytestmark = pytest.mark.gpu


def test_calls_on_two_streams_share_a_context():
    codec = hip.Codec(0)
    shape, counts, seed = synth.config("c2", 0.004)
    blob, off, nr, nh = synth.make_batches(shape, counts, seed)
    small_blob, small_off, small_nr, small_nh = synth.make_batches(*synth.config("c1", 0.001))
    big_blob, big_off, big_nr, big_nh = synth.make_batches(*synth.config("c1", 0.05))
    n_b = len(off) - 1
    d_blob, d_off = codec.to_device(blob, off)
    d_small, d_small_off = codec.to_device(small_blob, small_off)
    d_big, d_big_off = codec.to_device(big_blob, big_off)
    streams = [torch.cuda.Stream(device=codec.device) for _ in range(2)]
    cols = [hip.DeviceColumns(n_b, nr, max(nh, 1), codec.device) for _ in range(6)]
    small_cols = hip.DeviceColumns(len(small_off) - 1, small_nr, 1, codec.device)
    big_cols = hip.DeviceColumns(len(big_off) - 1, big_nr, 1, codec.device)
    torch.cuda.synchronize()
    fn = codec.lib.nisshi_hip_decode_dev

    def enqueue(d_b, length, d_o, n, c, stream):
        st = fn(codec.ctx, d_b.data_ptr(), length, d_o.data_ptr(), n, 1, ctypes.byref(c.c), c.rec_cap, c.hdr_cap,
                None, ctypes.c_void_p(stream.cuda_stream))
        assert st == C.OK

    # enqueue-only calls that alternate between two streams, a small call in between
    # (its workspace is the others'), then a call with more batches than any before:
    # the workspace grows while the earlier calls' kernels may still be queued
    for i, c in enumerate(cols):
        enqueue(d_blob, len(blob), d_off, n_b, c, streams[i % 2])
        if i == 2:
            enqueue(d_small, len(small_blob), d_small_off, len(small_off) - 1, small_cols, streams[1])
    enqueue(d_big, len(big_blob), d_big_off, len(big_off) - 1, big_cols, streams[0])
    torch.cuda.synchronize()
    st, want, _, _ = oracle.decode_columns(blob, off, True)
    for c in cols:
        assert_same(c.to_host(), want, n_b)
    st, want_small, _, _ = oracle.decode_columns(small_blob, small_off, True)
    assert_same(small_cols.to_host(), want_small, len(small_off) - 1)
    st, want_big, _, _ = oracle.decode_columns(big_blob, big_off, True)
    assert_same(big_cols.to_host(), want_big, len(big_off) - 1)
    codec.close()


def test_misaligned_columns_are_refused():
    codec = hip.Codec(0)
    blob, off, nr, nh = synth.make_batches(*synth.config("c1", 0.001))
    d_blob, d_off = codec.to_device(blob, off)
    cols = hip.DeviceColumns(len(off) - 1, nr, 1, codec.device)
    bad = C.CColumns()
    ctypes.memmove(ctypes.byref(bad), ctypes.byref(cols.c), ctypes.sizeof(bad))
    bad.rkey = cols.c.rkey + 8
    totals = hip.Totals()
    st = codec.lib.nisshi_hip_decode_dev(codec.ctx, d_blob.data_ptr(), len(blob), d_off.data_ptr(), len(off) - 1, 1,
                                         ctypes.byref(bad), nr, 1, ctypes.byref(totals), None)
    assert st == 9  # NISSHI_INVALID_ARGUMENT
    assert b"aligned" in codec.lib.nisshi_hip_last_error(codec.ctx)
    codec.close()
