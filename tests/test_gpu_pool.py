"""nisshi_hip_pool on a real GPU: two contexts on device 0 stand in for two
devices (an id may repeat), the frame's batches go to them by partition id, and
the columns come back in input order, equal to the CPU checker's decode of the
whole frame."""
import numpy as np
import pytest

import oracle
from tansu_amd import columns as C, hip
from test_pool import frame
from util import assert_same, offsets_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_frame_over_contexts_of_one_gpu(devices):
    batches, partition = frame()
    blob = np.frombuffer(b"".join(batches), dtype=np.uint8)
    off = offsets_of(batches)
    n_b = len(batches)
    st, want, nr, nh = oracle.decode_columns(blob, off)
    pool = hip.Pool(devices)
    try:
        for _ in range(2):   # the staging buffers are reused by the second call
            # (a batch that fails in its records keeps the record slots its header claimed)
            got, totals, per_device = pool.decode(blob, off, partition, len(blob) // 7 + 1, len(blob) + 1)
            assert totals.n_records >= nr and totals.n_failed == int((want["status"] != C.OK).sum())
            assert len(per_device) == len(devices) and sum(t.n_records for t in per_device) == totals.n_records
            assert_same(got, want, n_b)
            assert (got["data_off"][:n_b] == want["data_off"][:n_b]).all()
    finally:
        pool.close()
