"""Host-side synchronisation of the library (tostore_amd/csrc/tsh_host_sync.h), on the CPU: the handle
lock with asynchronous tickets and a waiting writer (pipelined submit + concurrent
append hung all three), the finalisation pool (every item once; parked when no call holds it), the
persistent shard workers of a multi-device handle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_sync_cpp(tmp_path):
    exe = tmp_path / "host_sync_test"
    src = os.path.join(ROOT, "tests", "cpp", "host_sync_test.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", str(exe), src], check=True)
    env = dict(os.environ, TSH_HOST_THREADS="6")
    env.pop("TSH_HOST_SPIN_US", None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=env)
    sys.stdout.write(r.stdout)
    sys.stderr.write(r.stderr)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("ok ") == 7
    # the library's group schedule (sharded_schedule) and bench.py's mirror of it, which the bench line reports
    sys.path.insert(0, ROOT)
    import bench

    n = 0
    for line in r.stdout.splitlines():
        if line.startswith("schedule ") or line.startswith("scheduleb "):
            head, _, sizes = line.partition(":")
            kind, nq, nbytes = head.split()
            rows = int(float(nbytes)) // (768 * 4)  # library_schedule takes rows x dim: bytes = rows * ld * 4
            assert bench.library_schedule(int(nq), rows, 768, kind == "scheduleb") == [int(x) for x in sizes.split()], line
            n += 1
    assert n == 2 * 17 * 7
