"""GPU: the library's own sharded search with MORE THAN ONE rank.  RCCL refuses two ranks on one device and a test
box has one GPU, so the ranks exchange over the host transport (tsh_comm_create_host + gloo): the protocol of
tsh_search_sharded -- groups, look-ahead scans on a helper thread, per-rank query slices, the result all-gather,
the overflow retry every rank takes alike, a failing rank staying in the collective -- runs exactly as it does over
RCCL, only the all-gather call differs.  And `python bench.py --gpus 2` as the driver would start it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_over_host_transport(hip_lib, world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           os.path.join(ROOT, "tests", "_comm_worker.py"), "40003"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-4000:]
    assert "MISMATCH" not in out, out[-4000:]
    assert out.count(" ok\n") == world * (2 * 8 + 1), out[-4000:]


def test_bench_gpus_2_plain_command(hip_lib):
    """Exactly what a driver without a launcher runs: two ranks share this box's GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ranks-share-gpu", "--backend",
                        "gloo", "--steps", "20", "--warmup", "5", "--rows", "200000", "--cpu-seconds", "3"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["recall_at_k"] == 1.0 and out["ids_and_distances_bit_exact"] is True and out["recall_queries"] >= 2
    assert "tsh_search_sharded" in out["config"]["sharding"]
    assert out["roofline"]["algorithmic_bytes_per_launch"] == 100000 * 768 * 4


def test_bench_in_process_multi_gpu(hip_lib):
    """`python bench.py --gpus 3 --in-process`: ONE process, one handle over three devices (here they share this box's GPU:
    --shards-share-gpu), host threads + copies + host merge, no ranks and no collective -- the same one-line contract."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--in-process", "--shards-share-gpu",
                        "--steps", "20", "--warmup", "5", "--rows", "200000", "--cpu-seconds", "3", "--recall-queries", "100"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["steps"] == 20 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["recall_at_k"] == 1.0 and out["ids_and_distances_bit_exact"] is True and out["recall_queries"] >= 100
    assert "IN ONE PROCESS" in out["config"]["sharding"] and "SHARE cuda:0" in out["config"]["sharding"]
    assert out["roofline"]["algorithmic_bytes_per_launch"] == 66688 * 768 * 4  # ceil(200 000 / 3) rounded up to whole tiles
    assert "exchange_timeline" not in out and "side" not in out


def test_bench_under_torch_distributed_run(hip_lib):
    """The driver's own launch line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- with two ranks sharing this box's
    GPU over the RCCL branch of the library (--fake-rccl): one JSON line from rank 0, with the exchange's account and the
    C4-per-rank leg in it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--fake-rccl", "--rows", "200000", "--c4-rows-per-rank", "50000", "--cpu-seconds", "3"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["recall_at_k"] == 1.0 and out["ids_and_distances_bit_exact"] is True
    assert "fake_rccl" in out["config"]["sharding"]
    tl = out["exchange_timeline"]
    assert [r["rank"] for r in tl["ranks"]] == [0, 1] and tl["ranks"][0]["calls"] > 0
    assert 0.8 <= tl["accounted_over_ms_per_step"]["rank0"] <= 1.2, tl["accounted_over_ms_per_step"]
    c4 = out["side"]["C4_per_rank"]
    assert c4["ids_and_distances_bit_exact"] is True and c4["batch_1024"]["value"] > 0 and c4["single_and_batched_agree"] is True
