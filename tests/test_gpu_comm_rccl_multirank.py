"""GPU: the RCCL branch of the library's sharded search with MORE THAN ONE rank.

Real RCCL refuses two ranks on one device and a test box has one GPU, so W processes share it and the library's
dlopen is pointed (TSH_RCCL_LIB) at tests/fake_rccl -- the same five entry points over shared memory, device buffers
bounced through the host on the caller's stream.  Everything of tsh_host_comm.inl.h's RCCL path then runs with real
pitches and slices: the d_all layout, the pitched hipMemcpy2DAsync of a rank's query slice, comm_agree's device path,
the device-side result all-gather, the overflow retry every rank takes alike, a failing rank staying in the
collective, a failing collective -- at W = 2, 3 and 8."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(world, args, extra_env=None, timeout=900):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_rccl

    lib = fake_rccl.build()
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, WORLD_SIZE=str(world), TSH_RCCL_LIB=lib, TSH_FAKE_RCCL_TIMEOUT_S="240",
                   OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs = [subprocess.Popen([sys.executable] + [a.replace("@TMP@", tmp) for a in args], cwd=ROOT,
                                  env=dict(env, RANK=str(r), LOCAL_RANK="0"), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=timeout)[0])
        finally:
            for p in procs:  # exactly the processes started above
                if p.poll() is None:
                    p.kill()
        return [p.returncode for p in procs], outs


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_search_over_the_rccl_branch(hip_lib, world):
    rcs, outs = run_ranks(world, [os.path.join(ROOT, "tests", "_rccl_worker.py"), "40003", "@TMP@/uid"])
    out = "\n".join(outs)
    assert all(rc == 0 for rc in rcs), out[-6000:]
    assert "MISMATCH" not in out, out[-6000:]
    assert out.count(" ok\n") == world * (3 * 12 + 2), out[-6000:]


@pytest.mark.parametrize("world", [2, 8])
def test_exchange_ahead_over_the_rccl_branch(hip_lib, world):
    """TSH_OPT_EXCHANGE_AHEAD = 1: every group's all-gather is launched when its scans are enqueued, stream-ordered
    behind their block writers; blocks carry their generation, blocks the host still has to redo (the overflow case of
    the worker) ask for a retry.  Same answers, same verdicts."""
    rcs, outs = run_ranks(world, [os.path.join(ROOT, "tests", "_rccl_worker.py"), "40003", "@TMP@/uid"],
                          {"WORKER_EXCHANGE_AHEAD": "1"})
    out = "\n".join(outs)
    assert all(rc == 0 for rc in rcs), out[-6000:]
    assert "MISMATCH" not in out, out[-6000:]
    assert out.count(" ok\n") == world * (3 * 12 + 2), out[-6000:]


def test_the_library_schedule_is_the_same_on_a_rank_without_a_handle(hip_lib):
    """The cut of a call into groups (10 + 5 + 5 for 20 queries on shards whose scan takes >= 30 us) may depend only on
    what the ranks agreed on: a rank whose handle is NULL knows neither rows nor dimension and must still enter the
    same three exchanges."""
    rcs, outs = run_ranks(2, [os.path.join(ROOT, "tests", "_rccl_worker.py"), "20011", "@TMP@/uid"],
                          {"WORKER_BIG_SHARDS": "1"})
    out = "\n".join(outs)
    assert all(rc == 0 for rc in rcs), out[-6000:]
    assert "MISMATCH" not in out and out.count("big shards") == 2 * 6, out[-3000:]
    assert "20 queries in 3 groups ok" in out
    assert "batching ranks: 20 queries in 1 group(s) ok" in out  # (round 6: a group is a batched call per shard)


def test_small_slots_chunk_the_stand_in(hip_lib):
    """The stand-in itself: transfers larger than its shared-memory slot go in pieces."""
    rcs, outs = run_ranks(2, [os.path.join(ROOT, "tests", "_rccl_worker.py"), "20011", "@TMP@/uid"],
                          {"TSH_FAKE_RCCL_SLOT": "4096"})
    out = "\n".join(outs)
    assert all(rc == 0 for rc in rcs) and "MISMATCH" not in out, out[-6000:]
