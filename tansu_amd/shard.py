"""Sharding and timing of a multi-process run: one process per GPU, each with its
own topic-partitions, no exchange on the data path.

Batches of different topic-partitions are independent (the reference keeps
nothing across them: nisshi-storage/src/service/produce.rs:165-191 loops over
partitions), so rank r of N simply works on its own shard.  The only
communication is the barrier around the timed region and the max-reduce of the
elapsed time; both go through torch.distributed over gloo, on the host.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

from . import synth


def shard_seed(seed: int, rank: int) -> int:
    """Seed of rank's shard: the same shape on every rank, different partitions."""
    return seed + 7919 * rank


C4_PARTITIONS = 1024        # the mixed workload: topic-partitions in all, 64 batches each
C4_PARTITIONS_PER_GPU = 128  # what a GPU of an eight-GPU node holds of them


def mixed_shard(scale: float, rank: int, world: int):
    """The mixed workload's partitions of one rank, laid out as one blob: partition
    p lives on device p mod 8 as on an eight-GPU node, and a job of fewer ranks
    holds the partitions of its first `world` devices (the same work a GPU, so
    that rates at 1, 2, 4 and 8 GPUs compare)."""
    import numpy as np
    per_gpu = max(5, int(C4_PARTITIONS_PER_GPU * scale))
    only = [rank + 8 * j for j in range(per_gpu)]
    parts = synth.mixed_partitions(C4_PARTITIONS, 64, only=only)
    blob = np.concatenate([p[0] for p in parts])
    off, base = [np.zeros(1, dtype=np.uint64)], 0
    for b, o, _, _ in parts:
        off.append(np.asarray(o[1:], dtype=np.uint64) + np.uint64(base))
        base += len(b)
    off = np.concatenate(off)
    return None, blob, off, sum(p[2] for p in parts), sum(p[3] for p in parts), len(off) - 1


def make_shard(workload: str, scale: float, rank: int, world: int = 1):
    """(shape, blob, batch_off, n_records, n_headers, n_batches) of one rank."""
    if workload == "c4":
        return mixed_shard(scale, rank, world)
    shape, counts, seed = synth.config(workload, scale)
    if workload == "c2v":
        blob, off, n_records, n_headers = synth.make_var_batches(len(counts), int(counts[0]), shard_seed(seed, rank))
        return shape, blob, off, n_records, n_headers, len(counts)
    blob, off, n_records, n_headers = synth.make_batches(shape, counts, shard_seed(seed, rank))
    return shape, blob, off, n_records, n_headers, len(counts)


def run_timed(step: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None],
              dist=None, device=None, on_start: Optional[Callable[[], None]] = None) -> float:
    """`warmup` untimed steps, then exactly `steps` timed ones bracketed by a
    barrier and a device sync on both sides; returns the MAX over ranks of the
    elapsed seconds.  `dist` is torch.distributed when the job has several ranks;
    `on_start` runs between the opening barrier and the first timed step."""
    import torch

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    for _ in range(warmup):
        step()
    barrier()
    if on_start is not None:
        on_start()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)   # a host tensor: the group is gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def job_rate(units_per_rank: int, world: int, steps: int, elapsed: float) -> float:
    """Whole-job throughput: what all ranks processed over the slowest rank's time."""
    return units_per_rank * world * steps / elapsed
