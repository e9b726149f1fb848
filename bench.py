#!/usr/bin/env python3
"""Device-resident RecordBatch v2 decode throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2]
    python bench.py --pool G            # one host frame over G devices (PCIe inclusive)

With --gpus N > 1 and no WORLD_SIZE in the environment, this process touches no
GPU: it starts N fresh processes of itself, one a rank (RANK, LOCAL_RANK,
WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT), passes rank 0's line on and
exits non-zero if any rank did.  Under `python -m torch.distributed.run` the ranks
are already there and each runs its share directly.

One step = one decode of the whole resident workload (batch headers, CRC-32C,
record walk into columns, header spans) through the C ABI.  At N > 1 every rank
decodes its own shard of topic-partitions (weak scaling, no collective on the
data path); the only communication is the barrier around the timed region and
the max-reduce of the elapsed time.

Prints one JSON line on rank 0.  `roofline` is for the kernel that takes most of
a step, with its duration measured by HIP events on the launch stream during the
timed steps.  `cpu_baseline` times the CPU restatement in oracle/ (a port of the
reference's algorithm, one core) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
EVENTS_EVERY = 4       # of the timed steps one in this many carries the two events around the dominant kernel

# Which input bytes and which columns each decode kernel is responsible for, as
# a fraction of the algorithmic bytes of a decode:
#   in_bytes + 64 * n_batches + 41 * n_records + 16 * n_headers
# `chunk_share`: the fraction of the input in batches taken chunk by chunk
# (chunk_kernel); `chunk_walks`: whether that kernel also decodes their records
# (batches of small records) or only takes their CRC.
def _walk_bytes(s, share):
    return share * (s["in_bytes"] + 41 * s["n_records"] + 16 * s["n_headers"])


KERNEL_BYTES = {
    "batch_header_kernel": lambda s: 64 * s["n_batches"],
    "chunk_kernel": lambda s: _walk_bytes(s, s["chunk_share"]) if s["chunk_walks"] else s["chunk_share"] * s["in_bytes"],
    "decode_fused_kernel": lambda s: _walk_bytes(s, 1 - s["chunk_share"]) if s["chunk_walks"]
    else _walk_bytes(s, 1.0) - s["chunk_share"] * s["in_bytes"],
    "second_pass_kernel": lambda s: 0,
}


def chunk_share(off, n_batches, n_records, in_bytes):
    """Which part of the input the library takes chunk by chunk, by its rule (batches
    of 512 KiB and more, of 128 KiB and more in calls of fewer than 6144 batches;
    the records with them when a record is small, here by the mean record)."""
    lens = np.diff(np.asarray(off, dtype=np.int64))
    chunks = (lens - 21 + 15 + 65535) // 65536
    big = chunks >= (2 if n_batches < 6144 else 8)
    walks = n_records > 0 and in_bytes / max(n_records, 1) <= 1024
    if not walks:
        big = chunks >= 8
    return float(lens[big].sum()) / max(in_bytes, 1), bool(walks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c2v", "c3", "c4", "c5", "c5g"])
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-copies", action="store_true",
                    help="also time the host-buffer entry point (PCIe inclusive), to stderr")
    ap.add_argument("--pool", type=int, default=0, metavar="G",
                    help="instead of the device-resident decode: one host frame of the mixed workload over G "
                         "devices through nisshi_hip_pool_decode, copies included (G of 1, 2, 4, 8)")
    args = ap.parse_args()

    from tansu_amd import shard
    if args.pool:
        return pool_bench(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no rank exists yet: start them, before anything here touches a GPU
        sys.exit(shard.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                    result_line=lambda line: line.startswith("{")))

    import torch
    from tansu_amd import columns as C, hip

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the job has {world} ranks", file=sys.stderr)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the ranks exchange nothing but the barrier around the timed region and the
        # elapsed time: host-side, over gloo (the data path has no collective)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    local %= max(1, torch.cuda.device_count())  # (more ranks than devices only when rehearsing on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    codec = hip.Codec(local)

    # ---- this rank's shard: its own topic-partitions, same shape, own seed
    shape, blob, off, n_records, n_headers, n_batches = shard.make_shard(args.workload, args.scale, rank, world)
    host_split = None
    if args.workload in ("c5", "c5g"):
        # C5 arrives as produce requests: the batches wrapped as ProduceRequest v11
        # partition data, found again on the host (one call over all records fields:
        # prefix decode and frame split) before the buffer goes to the device as it is
        import ctypes
        from tansu_amd import synth
        blob, field_pos, want_off = synth.make_produce_requests(blob, off, 7000 + rank)
        got_off = np.zeros(n_batches + 2, dtype=np.uint64)
        n_found = ctypes.c_uint64(0)
        lib = hip.load()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            st = lib.nisshi_records_fields_split(blob.ctypes.data, len(blob), field_pos.ctypes.data, len(field_pos), 1,
                                                 got_off.ctypes.data, len(got_off), None, ctypes.byref(n_found))
        host_split = {"ms": (time.perf_counter() - t0) / reps * 1e3, "records_fields": len(field_pos),
                      "what": "nisshi_records_fields_split over the request buffer, one host core"}
        assert st == 0 and n_found.value == n_batches and (got_off[:n_batches + 1] == want_off).all()
        off = got_off[:n_batches + 1].copy()
    in_bytes = len(blob)
    d_blob, d_off = codec.to_device(blob, off)
    cols = hip.DeviceColumns(n_batches, n_records, max(n_headers, 1), dev)
    # fill the CRC fields from the codec's own CRC (checked against the CPU
    # restatement in tests/ and in smoke()), so the decode sees valid batches
    codec.decode_dev(d_blob, in_bytes, d_off, n_batches, cols)
    crc = cols.t["crc_computed"][:n_batches].clone()
    starts = d_off[:n_batches] + 17
    for k in range(4):
        d_blob[starts + k] = ((crc >> (8 * (3 - k))) & 0xFF).to(torch.uint8)
    totals = codec.decode_dev(d_blob, in_bytes, d_off, n_batches, cols)
    assert (totals.n_records, totals.n_headers, totals.n_failed) == (n_records, n_headers, 0), \
        (totals.n_records, totals.n_headers, totals.n_failed)
    assert bool((cols.t["crc"][:n_batches] == cols.t["crc_computed"][:n_batches]).all())

    def step():
        codec.decode_dev(d_blob, in_bytes, d_off, n_batches, cols, sync=False)

    # The warm-up steps run with an event after every kernel: they give every kernel's
    # share of a step and say which kernel takes most of it.  The timed steps run
    # with events around that kernel only (an event between two kernels keeps them a
    # few microseconds apart, which a caller's decode does not pay), and only one
    # step in EVENTS_EVERY: that kernel's duration over those launches of the timed
    # region is what the roofline is computed from.
    state = {"all": {}, "dominant": "decode_fused_kernel"}

    def on_start():
        state["all"] = codec.kernel_times()
        if state["all"]:
            state["dominant"] = max(state["all"].items(), key=lambda kv: kv[1][0])[0]
        codec.profile(True, only=state["dominant"], every=EVENTS_EVERY)

    codec.profile(True)
    per_rank = []
    elapsed = shard.run_timed(step, args.steps, args.warmup, lambda: torch.cuda.synchronize(dev),
                              dist if world > 1 else None, dev, on_start=on_start, per_rank=per_rank)
    kt = codec.kernel_times()
    codec.profile(False)
    kt_all = state["all"] or kt

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    share, walks = chunk_share(off, n_batches, n_records, in_bytes)
    stats = dict(in_bytes=in_bytes, n_batches=n_batches, n_records=n_records, n_headers=n_headers,
                 chunk_share=share, chunk_walks=walks)
    step_ms = elapsed / args.steps * 1e3
    total_records = n_records * world
    algo_bytes = C.algorithmic_decode_bytes(n_batches, n_records, n_headers, in_bytes)
    name = state["dominant"]
    ms, launches = kt[name]
    avg_ms = ms / launches
    if "chunk_kernel" not in kt_all:
        # the chunks of large batches were worked on inside the fused kernel's launch
        # (calls of large batches): that kernel then has all the bytes of the decode
        stats["chunk_share"] = 0.0
    k_bytes = KERNEL_BYTES.get(name, lambda s: 0)(stats)
    achieved = k_bytes / (avg_ms * 1e-3) / 1e9
    # HBM-side bytes per launch of that kernel, from the counter passes of a committed
    # rocprofv3 profile of this workload (tools/profile_round.sh): reported only while
    # the kernel sources are the ones that profile was taken of
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        from tansu_amd import build as _build
        with open(tpath) as f:
            tj = json.load(f)
        src = tj.get("_source", {}).get(args.workload)
        if isinstance(src, dict):
            traffic_source = dict(src, csrc_sha_now=_build.source_hash())
            if src.get("csrc_sha") == traffic_source["csrc_sha_now"]:
                traffic = tj.get(args.workload, {}).get(name)
    result = {
        "metric": "device-resident RecordBatch v2 decode",
        "value": shard.job_rate(n_records, world, args.steps, elapsed),
        "unit": "records/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_ms,
        "ms_per_step_ranks": [t / args.steps * 1e3 for t in per_rank],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": {"c1": "C1: 10k batches x 100 records x 100 B values",
                         "c2": "C2: small records, 16 B keys, 64 B values, 2 headers, 1k records/batch",
                         "c2v": "C2 with sizes that change record to record: values 32-96 B, header values 0-32 B, "
                                "16 B keys, 2 headers, 1k records/batch",
                         "c3": "C3: large values, 1 MiB values, 4 records/batch",
                         "c4": "C4: mixed topic-partitions (five shapes by p mod 5, 64 batches each), 128 a GPU",
                         "c5": "C5: ragged batches, 1-10k records of 100 B, as ProduceRequest v11 partition data "
                               "(1-8 partitions a request, 1-4 batches a records field)",
                         "c5g": "C5g: C5 after a compaction that took every third record of every batch (offset deltas "
                                "0, 1, 3, 4, 6, ...), as ProduceRequest v11 partition data"}[args.workload],
            "batches_per_gpu": n_batches,
            "records_per_gpu": n_records,
            "input_bytes_per_gpu": in_bytes,
            "sharding": f"{world} x independent partitions, no collective",
        },
        "input_gbs": in_bytes * world * args.steps / elapsed / 1e9,
        "algorithmic_gbs": algo_bytes * world * args.steps / elapsed / 1e9,
        "roofline": {
            "bound": "hbm",
            "kernel": name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel_ms": avg_ms,
            "kernel_launches_timed": launches,
            "kernel_algorithmic_bytes": k_bytes,
        },
        # (from the warm-up steps, an event after every kernel)
        "kernels_ms": {k: v[0] / v[1] for k, v in sorted(kt_all.items(), key=lambda kv: -kv[1][0])},
    }
    if host_split:
        result["host_split"] = host_split

    # ---- the re-encode leg (deflated::Batch::try_from(inflated::Batch)), outside the
    # timed region: columns + payload -> wire bytes, which must equal the input
    out = torch.zeros(in_bytes + 64, dtype=torch.uint8, device=dev)
    out_off = torch.zeros(n_batches + 1, dtype=torch.int64, device=dev)
    te = codec.encode_dev(cols, n_batches, d_blob, cols.t["data_off"], out, in_bytes, out_off)
    if host_split is None:
        exact = te.out_bytes == in_bytes and bool(torch.equal(out[:in_bytes], d_blob[:in_bytes]))
    else:
        # the batches as they lie in the request buffer, without what is between them
        lens = cols.t["batch_length"][:n_batches].long() + 12
        src_idx = torch.repeat_interleave(d_off[:n_batches].long() - (torch.cumsum(lens, 0) - lens), lens) + \
            torch.arange(int(lens.sum()), device=dev)
        exact = te.out_bytes == int(lens.sum()) and bool(torch.equal(out[:te.out_bytes], d_blob[src_idx]))
    codec.profile(True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        codec.encode_dev(cols, n_batches, d_blob, cols.t["data_off"], out, in_bytes, out_off, sync=False)
    torch.cuda.synchronize(dev)
    enc_ms = (time.perf_counter() - t0) / args.steps * 1e3
    ekt = codec.kernel_times()
    codec.profile(False)
    payload = int((cols.t["rkey"][:n_records, 3].clamp(min=0).sum() + cols.t["rval"][:n_records, 1].clamp(min=0).sum()
                   + (cols.t["hdr"][:n_headers, 1].clamp(min=0).sum() + cols.t["hdr"][:n_headers, 3].clamp(min=0).sum()
                      if n_headers else 0)).item())
    enc_bytes = (algo_bytes - in_bytes) + payload + in_bytes   # columns and payload read, wire bytes written
    result["reencode"] = {
        "bit_exact": exact,
        "ms_per_step": enc_ms,
        "records_per_s": n_records / (enc_ms * 1e-3),
        "algorithmic_bytes": enc_bytes,
        "algorithmic_gbs": enc_bytes / (enc_ms * 1e-3) / 1e9,
        "roundtrip_records_per_s": n_records / ((step_ms + enc_ms) * 1e-3),
        "kernels_ms": {k: v[0] / v[1] for k, v in sorted(ekt.items(), key=lambda kv: -kv[1][0])},
    }

    # ---- what this device's memory system gives a plain copy (1 GiB read, 1 GiB
    # written), for reading `frac` against a measured number as well as the nominal peak
    result["roofline"]["device_copy_gbs"] = device_copy_gbs(dev)

    if args.with_copies:
        # the host-buffer entry point: H2D of the blob, the kernels, D2H of the columns;
        # once from ordinary host memory and once from page-locked buffers
        hdr_cap = max(n_headers, 1)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            codec.decode(blob, off, rec_cap=n_records, hdr_cap=hdr_cap)
        dt = (time.perf_counter() - t0) / reps
        p_blob = codec.pinned((in_bytes,), np.uint8)
        p_blob[:] = np.frombuffer(bytes(blob), dtype=np.uint8)
        p_off = np.ascontiguousarray(off, dtype=np.uint64)
        p_cols = codec.pinned_columns(n_batches, n_records, hdr_cap)
        codec.decode_into(p_blob, p_off, p_cols, n_records, hdr_cap)
        t0 = time.perf_counter()
        for _ in range(reps):
            tt = codec.decode_into(p_blob, p_off, p_cols, n_records, hdr_cap)
        dtp = (time.perf_counter() - t0) / reps
        assert tt.n_records == n_records and tt.n_failed == 0
        result["with_copies"] = {
            "pageable": {"records_per_s": n_records / dt, "input_gbs": in_bytes / dt / 1e9},
            "pinned": {"records_per_s": n_records / dtp, "input_gbs": in_bytes / dtp / 1e9,
                       "moved_gbs": (algo_bytes) / dtp / 1e9},
        }

    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(blob, off, n_batches, n_records)
        if args.workload != "c1":
            # C1 is the configuration the reference's own benchmark runs on a CPU: the same two
            # decodes over all of it (10 000 batches, 110 MB: beyond the caches), one core,
            # beside this workload's
            from tansu_amd import synth
            c1_blob, c1_off, c1_nr, _ = synth.make_batches(*synth.config("c1", 1.0))
            result["cpu_baseline_c1"] = cpu_baseline(c1_blob, c1_off, len(c1_off) - 1, c1_nr, budget_s=4.0,
                                                     all_cores=False)

    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def pool_bench(args):
    """One host frame of the mixed workload (C4's partitions of the first G devices
    of a node, in partition order) through nisshi_hip_pool_decode: the pool gathers
    each device's batches into page-locked staging, every device copies in, decodes
    and copies its columns out, side by side, and the pool puts the columns back in
    input order.  Everything is timed from host memory to host memory; on a box
    with fewer than G GPUs the device ids repeat (several contexts a GPU), which
    rehearses the scheduling but shares one GPU's PCIe link."""
    import torch
    from tansu_amd import columns as C, hip, shard
    g = args.pool
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise RuntimeError("no GPU visible: the HIP codec has no CPU fallback")
    blob, off, partition, n_records, n_headers = shard.pool_frame(args.scale, g)
    n_batches = len(off) - 1
    ids = [d % n_dev for d in range(g)]
    pool = hip.Pool(ids)
    # fill the CRC fields (a one-GPU decode of the same frame), so that the pool decodes valid batches
    codec = hip.Codec(0)
    d_blob, d_off = codec.to_device(blob, off)
    cols = hip.DeviceColumns(n_batches, n_records, max(n_headers, 1), codec.device)
    codec.decode_dev(d_blob, len(blob), d_off, n_batches, cols)
    from tansu_amd import synth
    synth.patch_crc(blob, off, cols.t["crc_computed"][:n_batches].cpu().numpy().view(np.uint32))
    d_blob, d_off = codec.to_device(blob, off)
    codec.decode_dev(d_blob, len(blob), d_off, n_batches, cols)
    want = cols.to_host(n_records, n_headers)
    del cols, d_blob, d_off
    hdr_cap = max(n_headers, 1) + 16 * g
    arrays = C.alloc_device_layout(n_batches, n_records, hdr_cap)
    blob = np.ascontiguousarray(blob)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    for _ in range(max(args.warmup, 1)):
        totals, per_device = pool.decode_into(blob, off, partition, arrays, n_records, hdr_cap)
    assert (totals.n_records, totals.n_headers, totals.n_failed) == (n_records, n_headers, 0), \
        (totals.n_records, totals.n_headers, totals.n_failed)
    # the pool's columns against one device's decode of the whole frame: batch for batch the same
    # records (a device's records are one run, so slots differ: compare through rec_base)
    got = C.unpack(arrays)
    for f in ("record_count", "crc", "crc_computed", "status", "batch_length", "base_offset"):
        assert (got[f][:n_batches] == want[f][:n_batches]).all(), f
    cnt = want["record_count"][:n_batches].astype(np.int64)
    idx_g = np.repeat(got["rec_base"][:n_batches].astype(np.int64), cnt) + \
        (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    idx_w = np.repeat(want["rec_base"][:n_batches].astype(np.int64), cnt) + \
        (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    for f in ("length", "offset_delta", "key_off", "key_len", "val_off", "val_len", "rhdr_count", "timestamp_delta", "rattributes"):
        assert (got[f][idx_g] == want[f][idx_w]).all(), f
    steps = max(1, min(args.steps, 10))
    times = []
    t0 = time.perf_counter()
    for _ in range(steps):
        pool.decode_into(blob, off, partition, arrays, n_records, hdr_cap)
        times.append(pool.last_times())
    elapsed = time.perf_counter() - t0
    pool.close()
    codec.close()
    mean = {k: sum(t[k] for t in times) / len(times) for k in times[0]}
    algo = C.algorithmic_decode_bytes(n_batches, n_records, n_headers, len(blob))
    print(json.dumps({
        "metric": "one host frame over G devices, host memory to host memory (nisshi_hip_pool_decode)",
        "value": n_records * steps / elapsed, "unit": "records/s", "n_gpus": len(set(ids)), "pool_devices": g,
        "device_ids": ids, "steps": steps, "warmup": max(args.warmup, 1), "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"C4 as one host frame: {n_batches} batches of {len(set(partition.tolist()))} "
                               f"partitions (five shapes by p mod 5, 64 batches each), device = partition mod {g}",
                   "input_bytes": int(len(blob)), "records": n_records, "headers": n_headers},
        "input_gbs": len(blob) * steps / elapsed / 1e9,
        "moved_gbs": algo * steps / elapsed / 1e9,
        "pool_ms": mean,
        "pool_ms_what": "host clock, the slowest device's: gather = its batches into page-locked staging; decode = "
                        "nisshi_hip_decode (copy in, kernels, copy out); scatter = its columns into the caller's, "
                        "header slots rebased; call = the whole nisshi_hip_pool_decode",
        "per_device_records": [int(t.n_records) for t in per_device],
        "checked": "columns equal one device's decode of the whole frame, batch for batch",
    }))


def device_copy_gbs(dev, n_bytes=1 << 30, reps=10):
    """Bytes read plus bytes written per second by a device-to-device copy."""
    import torch
    src = torch.empty(n_bytes // 16, 4, dtype=torch.int32, device=dev).fill_(1)
    dst = torch.empty_like(src)
    dst.copy_(src)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        dst.copy_(src)
    t1.record()
    torch.cuda.synchronize(dev)
    return 2 * n_bytes * reps / (t0.elapsed_time(t1) * 1e-3) / 1e9


def cpu_baseline(blob, off, n_batches, n_records, budget_s=12.0, all_cores=True):
    """The CPU restatement (oracle/, a port of the reference's algorithm) decoding
    a bounded sample of the same workload on one host core: the first batches of
    the workload, as many as fit the time budget, decoded repeatedly when the
    whole workload takes less than that.  `value` is the faithful form (CRC-32C,
    the reference's Vec<Record> with a Vec<Header> a record, then the columns);
    `lean` is the same decode straight into the columns, with no allocation a record:
    not the reference's way, but what a CPU decoder built for this output could do."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    probe = max(1, min(n_batches, 8))
    t0 = time.perf_counter()
    oracle.decode_columns(blob[:int(off[probe])], off[:probe + 1])
    per_batch = max(time.perf_counter() - t0, 1e-6) / probe
    sample = int(max(probe, min(n_batches, budget_s / per_batch)))
    sub = blob[:int(off[sample])]
    reps = int(max(1, min(64, budget_s / (per_batch * sample))))

    def timed(lean):
        total = 0
        t0 = time.perf_counter()
        for _ in range(reps):
            st, cols, nr, nh = oracle.decode_columns(sub, off[:sample + 1], lean=lean)
            assert st == 0 and (cols["status"] == 0).all()
            total += nr
        return total, nr, time.perf_counter() - t0
    total_records, nr, dt = timed(False)
    lean_records, _, lean_dt = timed(True)
    result = {
        "value": total_records / dt,
        "unit": "records/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {sample} of {n_batches} batches ({nr} records, {len(sub)} bytes) decoded {reps}x "
                  f"in {dt:.2f} s; C restatement of nisshi-sans-io (oracle/): CRC-32C, Vec<Record>-style "
                  "decode, then columns",
        "input_gbs": len(sub) * reps / dt / 1e9,
        "lean": {"value": lean_records / lean_dt, "unit": "records/s", "cores": 1,
                 "sample": f"the same sample decoded {reps}x in {lean_dt:.2f} s straight into the columns "
                           "(no Vec<Record>; same checks, same outcome)",
                 "input_gbs": len(sub) * reps / lean_dt / 1e9},
    }
    if not all_cores:
        return result
    # the same sample on the host cores this GPU's share of the box has (one
    # thread a core, each decoding its own run of batches; the C code runs
    # outside the interpreter lock)
    import concurrent.futures
    cores = max(1, min(16, len(os.sched_getaffinity(0)), sample))
    bounds = [sample * i // cores for i in range(cores + 1)]
    pieces = []
    for i in range(cores):
        lo, hi = bounds[i], bounds[i + 1]
        o = np.asarray(off[lo:hi + 1], dtype=np.uint64)
        pieces.append((blob[int(o[0]):int(o[-1])], o - o[0]))
    mreps = max(1, reps // 2)

    import ctypes
    from tansu_amd import columns as C

    def prepare(piece):
        # output columns allocated once, outside the timed region
        sub_blob, sub_off = piece
        n_b = len(sub_off) - 1
        rec_cap, hdr_cap = len(sub_blob) // 7 + 1, len(sub_blob) // 2 + 1
        cols = C.alloc_host(n_b, rec_cap, hdr_cap)
        return (np.ascontiguousarray(sub_blob), np.ascontiguousarray(sub_off, dtype=np.uint64), n_b, rec_cap,
                hdr_cap, cols, oracle.as_c(cols))

    def work(job):
        sub_blob, sub_off, n_b, rec_cap, hdr_cap, cols, cc = job
        n = 0
        r, h = ctypes.c_size_t(0), ctypes.c_size_t(0)
        for _ in range(mreps):
            st = oracle.LIB.no_decode_columns(sub_blob.ctypes.data, sub_off.ctypes.data, n_b, 1, ctypes.byref(cc),
                                              rec_cap, hdr_cap, ctypes.byref(r), ctypes.byref(h))
            assert st == 0
            n += r.value
        return n
    pieces = [prepare(p) for p in pieces]
    with concurrent.futures.ThreadPoolExecutor(cores) as pool:
        t0 = time.perf_counter()
        done = sum(pool.map(work, pieces))
        mdt = time.perf_counter() - t0
    result["all_cores"] = {"value": done / mdt, "unit": "records/s", "cores": cores,
                           "sample": f"the same sample split over {cores} threads, decoded {mreps}x in {mdt:.2f} s",
                           "input_gbs": len(sub) * mreps / mdt / 1e9}
    return result


if __name__ == "__main__":
    main()
