"""`bench.py --gpus N` with no ranks in the environment starts them itself
(tansu_amd/shard.py: launch_ranks): the environment every rank gets, whose output
is passed on, and what the job's exit code is when a rank fails or hangs."""
import io
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tansu_amd import shard  # noqa: E402


def script(tmp_path, body):
    path = tmp_path / "rank.py"
    path.write_text(textwrap.dedent(body))
    return [sys.executable, str(path)]


def test_rank_environment():
    env = shard.rank_env(3, 8, 29555, base={"PATH": "/bin", "WORLD_SIZE": "1"})
    assert env["RANK"] == env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29555" and env["PATH"] == "/bin"


def test_every_rank_runs_and_rank_zero_speaks(tmp_path):
    cmd = script(tmp_path, """
        import os, sys
        r, w = os.environ["RANK"], os.environ["WORLD_SIZE"]
        assert os.environ["LOCAL_RANK"] == r and os.environ["MASTER_ADDR"] == "127.0.0.1"
        open(os.path.join(sys.argv[1], "port" + r), "w").write(os.environ["MASTER_PORT"])
        print("line of rank", r, "of", w)
    """) + [str(tmp_path)]
    out, err = io.StringIO(), io.StringIO()
    assert shard.launch_ranks(cmd, 3, out=out, err=err) == 0
    assert out.getvalue() == "line of rank 0 of 3\n"
    assert sorted(err.getvalue().splitlines()) == ["line of rank 1 of 3", "line of rank 2 of 3"]
    ports = {(tmp_path / f"port{r}").read_text() for r in range(3)}
    assert len(ports) == 1 and int(ports.pop()) > 0


def test_a_failing_rank_fails_the_job_and_ends_the_others(tmp_path):
    cmd = script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)      # (as a rank waiting in a barrier for the one that died)
    """)
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    assert shard.launch_ranks(cmd, 2, out=out, err=err) == 7
    assert time.monotonic() - t0 < 30
    assert "rank 1 of 2 exited with 7" in err.getvalue()


def test_a_rank_that_never_ends_times_out(tmp_path):
    cmd = script(tmp_path, "import time; time.sleep(60)")
    out, err = io.StringIO(), io.StringIO()
    assert shard.launch_ranks(cmd, 2, timeout=1.0, out=out, err=err) == 124


def test_bench_starts_its_ranks_before_touching_a_gpu(tmp_path):
    """`bench.py --gpus 2` with no WORLD_SIZE: the parent launches two ranks.  Without a
    GPU every rank fails (the codec has no CPU fallback), so the job must fail too; with
    one, the line says n_gpus 2."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.is_available():
        args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c1", "--scale", "0.01",
                "--no-cpu-baseline"]
    else:
        args = ["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "c1", "--scale", "0.001",
                "--no-cpu-baseline"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=600)
    if torch.cuda.is_available():
        import json
        assert p.returncode == 0, p.stderr[-2000:]
        line = json.loads(p.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and len(line["ms_per_step_ranks"]) == 2
    else:
        assert p.returncode != 0
        assert "of 2 exited with" in p.stderr
