"""`bench.py --gpus 2` on a box with one GPU: the parent starts two ranks (both on
device 0), each decodes its own shard, and rank 0's line reports the whole job;
`bench.py --pool 2` runs one host frame over two contexts of the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_from_one_command():
    line = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "c1", "--scale", "0.05",
                      "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and len(line["ms_per_step_ranks"]) == 2
    assert line["ms_per_step"] == pytest.approx(max(line["ms_per_step_ranks"]))
    per_rank = line["config"]["records_per_gpu"]
    assert line["value"] == pytest.approx(2 * per_rank / (line["ms_per_step"] * 1e-3), rel=1e-6)
    assert line["reencode"]["bit_exact"] is True


def test_pool_leg():
    line = run_bench(["--pool", "2", "--scale", "0.08", "--steps", "2", "--warmup", "1"])
    assert line["pool_devices"] == 2 and len(line["per_device_records"]) == 2
    assert sum(line["per_device_records"]) == line["config"]["records"]
    assert line["pool_ms"]["call_ms"] > 0 and line["pool_ms"]["decode_ms"] > 0
