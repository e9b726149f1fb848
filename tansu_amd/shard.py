"""Sharding and timing of a multi-process run: one process per GPU, each with its
own topic-partitions, no exchange on the data path.

Batches of different topic-partitions are independent (the reference keeps
nothing across them: nisshi-storage/src/service/produce.rs:165-191 loops over
partitions), so rank r of N simply works on its own shard.  The only
communication is the barrier around the timed region and the max-reduce of the
elapsed time; both go through torch.distributed over gloo, on the host.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import tempfile
import time
from typing import Callable, List, Optional, Sequence

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


def pool_frame(scale: float, g: int):
    """One host frame of the mixed workload for a pool of `g` devices: the
    partitions the first `g` devices of an eight-GPU node hold (device d: d, d + 8,
    d + 16, ...), interleaved as a broker would receive them (partition order),
    with the partition id of every batch.  `g` must divide 8, so that
    `partition mod g` is the device that `partition mod 8` folds onto.
    Returns (blob, batch_off, partition, n_records, n_headers)."""
    import numpy as np
    assert 8 % g == 0, g
    per_gpu = max(5, int(C4_PARTITIONS_PER_GPU * scale))
    only = sorted(d + 8 * j for d in range(g) for j in range(per_gpu))
    parts = synth.mixed_partitions(C4_PARTITIONS, 64, only=only)
    blob = np.concatenate([p[0] for p in parts])
    off, partition, base = [np.zeros(1, dtype=np.uint64)], [], 0
    for pid, (b, o, _, _) in zip(only, parts):
        off.append(np.asarray(o[1:], dtype=np.uint64) + np.uint64(base))
        partition.append(np.full(len(o) - 1, pid, dtype=np.uint32))
        base += len(b)
    return (blob, np.concatenate(off), np.concatenate(partition), sum(p[2] for p in parts),
            sum(p[3] for p in parts))


def make_shard(workload: str, scale: float, rank: int, world: int = 1):
    """(shape, blob, batch_off, n_records, n_headers, n_batches) of one rank."""
    if workload == "c4":
        return mixed_shard(scale, rank, world)
    shape, counts, seed = synth.config(workload, scale)
    if workload == "c2v":
        blob, off, n_records, n_headers = synth.make_var_batches(len(counts), int(counts[0]), shard_seed(seed, rank))
        return shape, blob, off, n_records, n_headers, len(counts)
    blob, off, n_records, n_headers = synth.make_batches(shape, counts, shard_seed(seed, rank),
                                                         delta=synth.DELTAS.get(workload))
    return shape, blob, off, n_records, n_headers, len(counts)


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    """The environment of rank `rank` of a `world`-rank job on this node: what
    `torch.distributed.run` would set, rendezvous on the loopback address."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    return env


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(cmd: Sequence[str], world: int, timeout: Optional[float] = None, out=None, err=None,
                 result_line: Optional[Callable[[str], bool]] = None) -> int:
    """Run `cmd` as `world` fresh processes, one a rank, each with the environment
    of `rank_env`, and wait for them.  The caller must not have touched the GPU:
    every rank is a child of its own, started before any device call is made here.
    Rank 0's standard output is passed on (`out`, default this process's), the
    other ranks' goes to `err` so that the job prints ONE result line (with
    `result_line`, only rank 0's lines for which it holds are passed on to `out`:
    libraries print notices of their own to standard output).  Returns 0
    when every rank did; otherwise the first non-zero exit code seen, after the
    ranks still running were ended (by their own process ids); 124 on `timeout`."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port()
    procs: List[subprocess.Popen] = []
    outs = [tempfile.TemporaryFile(mode="w+") for _ in range(world)]   # (a file, not a pipe: no rank can block on it)
    for rank in range(world):
        procs.append(subprocess.Popen(list(cmd), env=rank_env(rank, world, port), stdout=outs[rank], stderr=None))
    deadline = None if timeout is None else time.monotonic() + timeout
    code = 0
    pending = set(range(world))
    while pending and code == 0:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                code = rc if rc > 0 else 128 - rc
                print(f"rank {r} of {world} exited with {rc}", file=err)
                break
        if deadline is not None and time.monotonic() > deadline and pending and code == 0:
            code = 124
            print(f"ranks {sorted(pending)} still running after {timeout} s", file=err)
        if pending and code == 0:
            time.sleep(0.05)
    for r in sorted(pending):           # a rank failed (or time ran out): the others wait for it in a barrier
        procs[r].terminate()
    for r in sorted(pending):
        try:
            procs[r].wait(10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    for r, f in enumerate(outs):
        f.seek(0)
        text = f.read()
        f.close()
        for line in text.splitlines(keepends=True):
            mine = r == 0 and (result_line is None or result_line(line))
            (out if mine else err).write(line)
    out.flush()
    return code


def run_timed(step: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None],
              dist=None, device=None, on_start: Optional[Callable[[], None]] = None,
              per_rank: Optional[list] = None) -> float:
    """`warmup` untimed steps, then exactly `steps` timed ones bracketed by a
    barrier and a device sync on both sides; returns the MAX over ranks of the
    elapsed seconds.  `dist` is torch.distributed when the job has several ranks;
    `on_start` runs between the opening barrier and the first timed step;
    `per_rank`, when given, is filled with every rank's own elapsed seconds."""
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
    mine = elapsed
    if dist is not None:
        world = dist.get_world_size()
        t = torch.zeros(world, dtype=torch.float64)   # a host tensor: the group is gloo
        t[dist.get_rank()] = mine
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(t.max().item())
        if per_rank is not None:
            per_rank[:] = [float(x) for x in t.tolist()]
    elif per_rank is not None:
        per_rank[:] = [mine]
    return elapsed


def job_rate(units_per_rank: int, world: int, steps: int, elapsed: float) -> float:
    """Whole-job throughput: what all ranks processed over the slowest rank's time."""
    return units_per_rank * world * steps / elapsed
