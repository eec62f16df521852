"""CPU: `python bench.py --gpus N` started plainly (no torch.distributed.run, no WORLD_SIZE) must start its own
N ranks, keep the one-JSON-line contract on stdout and hand a failing rank's return code through (round 2's
bench answered `--gpus 2` with SystemExit: a driver with an 8-GPU node would have got no scaling point at all).
The ranks run bench.py's real flow over gloo; only the device side is the oracle-backed stand-in of
tests/fake_bench_env.py (TSH_BENCH_ENV)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--rows", "3000", "--dim", "32", "--k", "10", "--cpu-seconds", "0.2", "--steps", "20", "--warmup", "5",
         "--c4-rows-per-rank", "200"]


def _run(extra_env, *args, timeout=300):
    env = dict(os.environ, TSH_BENCH_ENV="fake_bench_env:FakeEnv", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, *args], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_plain_command_starts_its_own_ranks():
    p = _run({}, "--gpus", "2")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["value"] > 0
    assert out["scaling"] == "strong" and out["config"]["workload"].startswith("C2")
    assert out["recall_at_k"] == 1.0 and out["ids_and_distances_bit_exact"] is True
    assert out["roofline"]["algorithmic_bytes_per_launch"] == 1500 * 32 * 4  # rank 0's shard
    assert "cpu_baseline" not in out  # N = 1 only
    # both ranks' accounts reach rank 0's line; the weak-scaling leg (C4's per-rank shape) ran on both
    assert [r["rank"] for r in out["exchange_timeline"]["ranks"]] == [0, 1]
    assert len(out["host_cpu"]["per_rank"]) == 2
    c4 = out["side"]["C4_per_rank"]
    assert c4["n_gpus"] == 2 and "400x1536" in c4["workload"] and c4["ids_and_distances_bit_exact"] is True


def test_a_failing_rank_fails_the_command():
    p = _run({"TSH_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--exchange", "torch")
    assert p.returncode != 0
    assert p.stdout.strip() == "", p.stdout
    assert "told to fail" in p.stderr


def test_launch_timeout_gives_up():
    # rank 1 never reaches the first collective: the launcher must end the job instead of waiting forever
    p = _run({"TSH_BENCH_HANG_RANK": "1"}, "--gpus", "2", "--exchange", "torch", "--launch-timeout", "20")
    assert p.returncode == 124, (p.returncode, p.stderr[-2000:])
    assert p.stdout.strip() == ""
