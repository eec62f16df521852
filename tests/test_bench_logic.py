"""bench.py without a GPU: the whole flow of run_bench() -- query pool, timed regions, group
arithmetic, cpu_baseline / recall legs, side legs, JSON line -- driven through a CPU stand-in for
the device side (the stand-in answers searches with the oracle; this file is test code, the
product path never does that).  Round 1's bench died on `queries[25]` with the driver's own
arguments (--steps 20 --warmup 5): every (steps, warmup) pair the driver or a profile run uses
is exercised here, for N = 1 and for the N > 1 branch."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_bench_env import FakeEnv  # noqa: E402


def _args(*extra):
    return bench.parse(["--rows", "3000", "--dim", "32", "--k", "10", "--cpu-seconds", "0.2",
                        "--recall-queries", "150", "--recall-seconds", "5", "--c4-rows-per-rank", "300",
                        "--c3-check", "200", "--c5-check", "40"] + list(extra))


DRIVER_ARGS = [(1, 0), (5, 0), (20, 5), (1000, 50)]


@pytest.mark.parametrize("steps,warmup", DRIVER_ARGS)
def test_single_gpu_line(oracle_mod, steps, warmup):
    a = _args("--steps", str(steps), "--warmup", str(warmup), "--side", "c5,c1" if steps != 20 else "c5,c1,c3")
    env = FakeEnv(oracle_mod)
    line = bench.run_bench(a, env)
    out = json.loads(line)  # strict JSON: a bare NaN would fail here
    assert "NaN" not in line and "Infinity" not in line
    assert out["steps"] == steps and out["warmup"] == warmup and out["n_gpus"] == 1
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert out["timed_regions"]["count"] == len(out["timed_regions"]["seconds"]) == bench.auto_repeats(steps)
    assert out["unit"] == "queries/s" and out["dtype"] == "f32" and out["vs_baseline"] is None
    assert out["config"]["workload"].startswith("C2")
    assert out["config"]["queries_in_flight"] == min(8, 64, steps)
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["algorithmic_bytes_per_launch"] == 3000 * 32 * 4
    assert out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    assert out["recall_at_k"] == 1.0 and out["ids_and_distances_bit_exact"] is True
    assert out["recall_queries"] == 150
    side = out["side"]
    for keep in ("keep_1%", "keep_10%", "keep_50%", "keep_100%", "keep_10%_range"):  # SURVEY section 8d's list
        assert side["C5"][keep]["ids_and_distances_bit_exact"] is True, side["C5"]
        assert side["C5"][keep]["checked_queries"] == 40
        assert side["C5"][keep]["library_default_path"]["ids_and_distances_bit_exact"] is True
        mh = side["C5"][keep]["mask_handle"]  # the same mask as a device-resident handle: pipelined and one at a time
        assert mh["ids_and_distances_bit_exact"] is True and mh["value"] > 0
        assert set(mh["one_at_a_time_us"]) == {"pointer", "handle"} and mh["one_at_a_time_us"]["handle"]["p50"] > 0
    assert side["C5"]["keep_10%_range"]["mask"] == "range" and side["C5"]["keep_100%"]["kept_rows"] == 3000
    assert side["C1"]["ids_and_distances_bit_exact"] is True and side["C1"]["latency_us"]["p50"] > 0
    ann = side["C1"]["reference_ann_restated"]  # context only, labelled as a restatement
    assert "error" not in ann and 0.0 <= ann["recall_at_k"] <= 1.0 and ann["ms_per_query"] > 0
    assert "restatement" in ann["label"] and "unverifiable" in ann["label"]
    if steps == 20:
        c3 = side["C3"]
        assert c3["ids_and_distances_bit_exact"] is True and c3["checked_queries"] == 200
        assert c3["ms_per_step"] <= c3["ms_per_step_p99"] <= c3["ms_per_step_max"]
        assert c3["roofline"]["bound"] == "mfma" and c3["f32_mfma_variant"]["roofline"]["peak"] == 157.3
        assert set(c3["smaller_calls"]) == {"16_queries", "128_queries"}
    assert all(i.closed for i in env.made)


def test_c4_shard_of_8_leg(oracle_mod):
    """side.C4_shard_of_8: one rank's share of BASELINE.json's C4 in the N = 1 line -- the
    sharded entry point over a world of one, the driver's regions, the oracle over that shard (streamed: no host
    copy), and the 1024-query call on the shard's matrix cores."""
    a = _args("--steps", "20", "--warmup", "5", "--side", "c4s8")
    env = FakeEnv(oracle_mod)
    out = json.loads(bench.run_bench(a, env))
    assert set(out["side"]) == {"C4_shard_of_8", "seconds"}
    c4 = out["side"]["C4_shard_of_8"]
    assert "error" not in c4, c4
    assert c4["workload"].startswith("one rank's share of C4 at N = 8: 300x1536 f32, ip, k=100, 20 single-query steps")
    assert c4["roofline"]["algorithmic_bytes_per_launch"] == 300 * 1536 * 4 and c4["upper_bound_speedup"] is None
    assert c4["timed_regions"]["count"] == bench.auto_repeats(20) and c4["exchange_timeline"]["ranks"][0]["calls"] >= 3
    assert c4["ids_and_distances_bit_exact"] is True and c4["recall_at_k"] == 1.0 and c4["checked_queries"] == 16
    b = c4["batch_1024"]
    assert "error" not in b and b["value"] > 0 and b["single_and_batched_agree"] is True and b["roofline"]["bound"] == "mfma"
    assert all(i.closed for i in env.made)


@pytest.mark.parametrize("steps,warmup", DRIVER_ARGS)
def test_sharded_branch_arithmetic(oracle_mod, steps, warmup):
    """rank 0 of a two-rank job (collectives stubbed): group sizes, n_cpu clamp, JSON."""
    a = _args("--steps", str(steps), "--warmup", str(warmup), "--gpus", "2")
    env = FakeEnv(oracle_mod, world=2)
    out = json.loads(bench.run_bench(a, env))
    assert out["n_gpus"] == 2 and out["value"] > 0
    # the N > 1 line says where the time went, phase by phase, against the step
    tl = out["exchange_timeline"]
    assert len(tl["ranks"]) == 1 and tl["ranks"][0]["calls"] == bench.auto_repeats(steps)
    assert abs(tl["phases_sum_ms_per_step"] - 0.99 * tl["call_ms_per_step"]) < 1e-6 * max(1.0, tl["call_ms_per_step"])
    # (the account is a MEAN over all regions, ms_per_step the MEDIAN region: with one-step regions of microseconds the
    # ratio may pass 1)
    assert 0.0 < tl["phases_sum_over_ms_per_step"] <= 3.0
    assert tl["ranks"][0]["host_cpu"]["cpus_busy"] > 0
    # and carries the weak-scaling point: BASELINE.json C4's per-rank shape (here 300 rows per rank)
    c4 = out["side"]["C4_per_rank"]
    assert c4["scaling"] == "weak" and c4["n_gpus"] == 2 and "600x1536" in c4["workload"] and c4["value"] > 0
    assert c4["ids_and_distances_bit_exact"] is True and c4["single_and_batched_agree"] is True
    assert c4["batch_1024"]["value"] > 0 and c4["exchange_timeline"]["ranks"][0]["calls"] >= 3
    assert c4["roofline"]["algorithmic_bytes_per_launch"] == 300 * 1536 * 4
    assert out["roofline"]["algorithmic_bytes_per_launch"] == 1500 * 32 * 4
    assert out["recall_at_k"] == 1.0 and out["ids_and_distances_bit_exact"] is True
    g = bench.sharded_group(64, steps)
    assert out["config"]["queries_per_call"] == g >= 1
    timed = [x for x in env.searchers[0].groups if x[0] == steps]
    assert len(timed) == bench.auto_repeats(steps) and all(x[1] == g for x in timed)


@pytest.mark.parametrize("extra", [["--inflight", "1"], ["--group", "0", "--inflight", "3"],
                                   ["--mask-keep", "0.1"], ["--mask-keep", "0.5", "--mask-kind", "range"],
                                   ["--no-cpu-baseline"], ["--metric", "cosine"], ["--metric", "ip"]])
def test_option_paths(oracle_mod, extra):
    a = _args("--steps", "7", "--warmup", "2", "--no-side", *extra)
    out = json.loads(bench.run_bench(a, FakeEnv(oracle_mod)))
    assert out["value"] > 0
    if "--no-cpu-baseline" in extra:
        assert "cpu_baseline" not in out
    else:
        assert out["ids_and_distances_bit_exact"] is True and out["recall_at_k"] == 1.0
    if extra[:2] == ["--inflight", "1"]:
        assert out["config"]["queries_in_flight"] == 1
    if extra[:2] == ["--group", "0"]:
        assert out["config"]["queries_in_flight"] == 3


def test_batch_main_line(oracle_mod):
    a = _args("--steps", "20", "--warmup", "5", "--batch", "64", "--metric", "cosine")
    out = json.loads(bench.run_bench(a, FakeEnv(oracle_mod)))
    assert out["config"]["workload"].startswith("C3") and out["ids_and_distances_bit_exact"] is True
    assert out["roofline"]["bound"] == "mfma"


def test_sizing_helpers():
    for steps, warmup in DRIVER_ARGS:
        pool = bench.query_pool_size(steps, warmup, 1000)
        assert pool >= warmup + steps and pool >= 64 and pool >= 1000
        assert 3 <= bench.auto_repeats(steps) <= 25
        assert bench.sharded_group(64, steps) >= 1
    assert bench.auto_repeats(20) == 25 and bench.auto_repeats(1000) == 3 and bench.auto_repeats(5, 4) == 4
    # the round-1 crash: a fast host wants 32 baseline queries, the pool had 25
    import bench_check

    assert bench_check.cpu_sample_size(0.1, 15.0, 25) == 25
    assert bench_check.cpu_sample_size(0.47, 15.0, 1000) == 31
    assert bench_check.cpu_sample_size(100.0, 15.0, 1000) == 2
    assert bench_check.cpu_sample_size(100.0, 15.0, 1) == 1
    line = bench.dumps({"a": float("nan"), "b": [np.float32(1.5), float("inf")], "c": np.int64(3), "d": np.bool_(True)})
    assert json.loads(line) == {"a": None, "b": [1.5, None], "c": 3, "d": True}


def test_the_oracle_is_out_of_reach_of_timed_code(oracle_mod):
    """bench.py reaches oracle/ through bench_check.py only, and bench_check refuses to run (or load) while a
    timed region is open."""
    import ast
    import importlib

    import bench_check

    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            assert all(al.name.split(".")[0] != "oracle" for al in node.names), "bench.py imports oracle"
        if isinstance(node, ast.ImportFrom):
            assert (node.module or "").split(".")[0] != "oracle", "bench.py imports from oracle"
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name):
            assert node.value.id != "oracle", "bench.py calls into oracle directly (line %d)" % node.lineno
    rows = np.zeros((8, 4), np.float32)
    with bench.timed_region():
        with pytest.raises(RuntimeError, match="timed region"):
            bench_check.oracle_answers(rows, rows[:1], 0, 2)
        with pytest.raises(RuntimeError, match="timed region"):
            importlib.reload(bench_check)
    assert os.environ.get("TSH_BENCH_TIMED") is None
    importlib.reload(bench_check)
    assert bench_check.oracle_answers(rows, rows[:1], 0, 2)[2][0] == 2


def test_shard_of_8_leg(oracle_mod):
    """side.shard_of_8: one rank's share of the headline at N = 8 through the sharded entry
    point, in the driver's shape -- regions of --steps queries, fenced -- with its exchange_timeline and the speed-up it
    bounds."""
    a = _args("--steps", "20", "--warmup", "5", "--side", "s8")
    env = FakeEnv(oracle_mod)
    out = json.loads(bench.run_bench(a, env))
    s8 = out["side"]["shard_of_8"]
    assert "error" not in s8, s8
    assert s8["steps"] == 20 and s8["us_per_query"] > 0 and abs(s8["ms_per_step"] * 1e3 - s8["us_per_query"]) < 1e-9
    assert abs(s8["upper_bound_speedup"] - out["ms_per_step"] / s8["ms_per_step"]) < 1e-9
    assert s8["timed_regions"]["count"] == len(s8["timed_regions"]["seconds"]) == bench.auto_repeats(20)
    assert s8["workload"].startswith("one rank's share of C2 at N = 8: 375x32")  # rows / 8 of the test's 3000 x 32 corpus
    assert s8["roofline"]["algorithmic_bytes_per_launch"] == 375 * 32 * 4
    assert s8["exchange_timeline"]["ranks"][0]["calls"] == bench.auto_repeats(20)
    assert s8["ids_and_distances_bit_exact"] is True and s8["recall_at_k"] == 1.0 and s8["checked_queries"] == 16
    assert sum(s8["queries_per_exchange"]) == 20 and set(s8["group_sweep_us_per_query"]) == {"1", "2", "4", "5", "10", "20"}
    assert bench.library_schedule(20, 125_000, 768) == [10, 5, 5] and bench.library_schedule(20, 10_000, 128) == [20]
    assert bench.library_schedule(300, 125_000, 768) == [256, 44] and bench.library_schedule(1, 1, 1) == [1]
    assert bench.library_schedule(20, 125_000, 768, batched=True) == [20] and bench.library_schedule(1500, 125_000, 768) == [512, 512, 476]
    assert all(i.closed for i in env.made)


@pytest.mark.parametrize("n_dev", [2, 8])
def test_in_process_multi_gpu_line(oracle_mod, n_dev):
    """`bench.py --gpus N --in-process`: ONE process, one handle over N devices, no ranks and no
    collective -- the same line contract, n_gpus = N, the per-device shard in the roofline, config.sharding saying which
    exchange ran; the launcher is not involved."""
    a = _args("--steps", "20", "--warmup", "5", "--gpus", str(n_dev), "--in-process")
    env = FakeEnv(oracle_mod)
    out = json.loads(bench.run_bench(a, env))
    assert env.n_devices_seen == n_dev and env.world == 1
    assert out["n_gpus"] == n_dev and out["value"] > 0 and out["scaling"] == "strong" and "side" not in out
    assert "IN ONE PROCESS" in out["config"]["sharding"] and "no collective" in out["config"]["sharding"]
    per = ((3000 + n_dev - 1) // n_dev + 63) // 64 * 64
    assert out["roofline"]["algorithmic_bytes_per_launch"] == min(per, 3000) * 32 * 4 and out["roofline"]["traffic"] is None
    assert out["recall_at_k"] == 1.0 and out["ids_and_distances_bit_exact"] is True and out["cpu_baseline"]["cores"] == 1
    assert "exchange_timeline" not in out
    assert all(i.closed for i in env.made)
