"""The multi-process path of bench.py on CPU: two ranks over gloo, each with its
own shard, no exchange on the data path; the job rate is everything both ranks
processed over the slower rank's time.  The CPU restatement stands in for the
device codec here (this is a test of the sharding and timing logic only)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _rank_main(rank, world, port, scale, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tansu_amd import shard
    import oracle
    shape, blob, off, n_records, n_headers, n_batches = shard.make_shard("c1", scale, rank)
    seen = {"steps": 0, "records": 0}

    def step():
        st, cols, nr, nh = oracle.decode_columns(blob, off)
        assert st == 0 and nr == n_records
        seen["steps"] += 1
        seen["records"] += nr
        if rank == 1:  # the slower rank sets the job's time
            import time
            time.sleep(0.02)

    started = {"at": None}
    elapsed = shard.run_timed(step, steps=3, warmup=1, sync=lambda: None, dist=dist, device=None,
                              on_start=lambda: started.__setitem__("at", seen["steps"]))
    digest = int(np.frombuffer(bytes(blob[:4096]), dtype=np.uint8).astype(np.uint64).sum())
    out[rank] = (elapsed, seen["steps"], started["at"], n_records, digest,
                 shard.job_rate(n_records, world, 3, elapsed))
    dist.destroy_process_group()


def test_two_ranks_shard_and_time_together():
    world, port = 2, 29000 + os.getpid() % 2000
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_rank_main, args=(world, port, 0.002, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    # both ranks report the same (max) time and ran warmup + timed steps
    assert r0[0] == pytest.approx(r1[0]) and r0[0] >= 3 * 0.02
    assert r0[1] == r1[1] == 4 and r0[2] == r1[2] == 1
    # same shape, different partitions
    assert r0[3] == r1[3] and r0[4] != r1[4]
    # whole-job rate = both shards over the common time
    assert r0[5] == pytest.approx(2 * r0[3] * 3 / r0[0])


def test_shard_seeds_differ():
    from tansu_amd import shard
    assert len({shard.shard_seed(1, r) for r in range(8)}) == 8


def test_mixed_workload_ranks_hold_their_own_partitions():
    """The mixed workload: partition p lives on device p mod 8, so the ranks of a
    job hold different partitions of the same five shapes, the same work each."""
    from tansu_amd import shard
    a = shard.make_shard("c4", 10 / 128, 0, 2)
    b = shard.make_shard("c4", 10 / 128, 1, 2)
    assert a[5] == b[5] == 640
    assert bytes(a[1][:4096]) != bytes(b[1][:4096])
    # rank 0 holds partitions 0, 8, 16, ...: shapes (p mod 5) 0, 3, 1, 4, 2, ...; rank 1: 1, 4, 2, 0, 3, ...
    sizes_a = np.diff(a[2].astype(np.int64)).reshape(10, 64)[:, 0]
    sizes_b = np.diff(b[2].astype(np.int64)).reshape(10, 64)[:, 0]
    assert sorted(sizes_a[:5]) == sorted(sizes_b[:5]) and list(sizes_a[:5]) != list(sizes_b[:5])
