"""`python bench.py --gpus N` must start by itself: with no launcher in the environment the
parent spawns its N ranks before any GPU call, forwards rank 0's JSON line as the LAST line of stdout, exits
non-zero when a rank does, and never waits forever on a dead or hung peer.  CPU-only: the ranks run bench.py's
launcher rehearsal (gloo rendezvous + the reductions `measure` ends with) — the product itself has no CPU path;
the same self-launch with the real workload runs on the GPU box in tests/test_gpu_rccl.py."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra_env=None, timeout=120, args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"DGMI_BENCH_REHEARSAL": "launcher", "DGMI_SKIP_BUILD": "1"})
    env.update(extra_env or {})
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", *args],
                       capture_output=True, text=True, timeout=timeout, env=env)
    return r, time.monotonic() - t0


def test_self_launch_world2_forwards_rank0_json_last():
    r, _ = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    line = json.loads(lines[-1])  # the JSON line is the LAST line of stdout
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["max_over_ranks"] == 2.0
    assert line["steps"] == 2 and line["warmup"] == 1  # the flags reached the ranks
    assert sum(l.lstrip().startswith("{") for l in lines) == 1  # printed once
    assert any("rendezvous at 127.0.0.1:" in l for l in lines)  # rank 0's other output is relayed too


def test_self_launch_world4():
    r, _ = _run(4)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["ranks_seen"] == 4


def test_a_failing_rank_ends_the_job_non_zero_and_promptly():
    r, took = _run(2, {"DGMI_BENCH_REHEARSAL_FAIL_RANK": "1"})
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with 3" in r.stderr
    assert took < 60  # rank 0 was blocked in the rendezvous: it was stopped, not waited for


def test_a_hung_rank_is_stopped_at_the_time_limit():
    r, took = _run(2, {"DGMI_BENCH_REHEARSAL_HANG_RANK": "1", "DGMI_BENCH_LAUNCH_TIMEOUT": "8"})
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert "still running" in r.stderr and took < 60


def test_under_a_launcher_the_world_size_must_match():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", DGMI_SKIP_BUILD="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in (r.stderr + r.stdout)
