"""GPU: the one-process-per-GPU code path end to end with real processes:
torch.distributed.run starts 3 ranks that share this box's single GPU (gloo
exchange), each owning a row range; ShardedSearcher.search and .search_many
must equal the oracle over the whole corpus on every rank."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_three_ranks_one_gpu(hip_lib):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_multirank_worker.py"), "50001"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert "MISMATCH" not in out, out[-3000:]
    assert out.count(" ok ") == 6, out[-3000:]  # 3 ranks x (search, search_many)
