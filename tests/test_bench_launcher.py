"""bench.py started plainly with --gpus N spawns its N ranks itself (before anything touches a GPU), rendezvous on 127.0.0.1, and prints
rank 0's ONE JSON line.  Checked here without a GPU: --dry-run keeps the launcher, the gloo process group, the barriers and the
max-over-ranks reduction, and skips the kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_dry(n, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run", "--dist-backend", "gloo", "--steps", "3", *extra],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    line = run_dry(2)
    assert line["dry_run"] and line["n_gpus"] == 2 and line["steps"] == 3
    assert line["rank_id_sum"] == 1.0            # ranks 0 and 1 both took part in the collective
    assert line["max_rank_seconds"] >= 0.02      # the reported time is the slowest rank's (rank 1 sleeps 20 ms)


def test_self_launch_three_ranks_lsi_flag_travels():
    line = run_dry(3, ("--workload", "lsi"))
    assert line["n_gpus"] == 3 and line["rank_id_sum"] == 3.0 and line["workload"] == "lsi"


def test_single_rank_needs_no_launcher():
    line = run_dry(1)
    assert line["n_gpus"] == 1 and line["rank_id_sum"] == 0.0


def test_a_rank_that_dies_takes_the_launcher_down():
    """a rank that fails before the rendezvous must not leave the others waiting for their timeout: the launcher stops them and reports
    the failing rank's code and stderr"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LEXLS_BENCH_FAIL_RANK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--dist-backend", "gloo", "--steps", "3"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 7
    assert "rank 1 exited with code 7" in out.stderr
