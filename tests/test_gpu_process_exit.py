"""GPU: a host process that used the library exits cleanly -- also under rocprofv3, where CU-masked streams left to
the runtime's own teardown used to crash python at exit (after the profile was written)."""
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
import sys
import numpy as np
sys.path.insert(0, %r)
from tostore_amd import HipVectorIndex
idx = HipVectorIndex(64, 0)
idx.append(0, np.random.default_rng(1).standard_normal((5000, 64)).astype(np.float32))
ids, dist, cnt = idx.search(np.zeros(64, np.float32), 5)
assert cnt[0] == 5
print("done", flush=True)
""" % ROOT


@pytest.mark.parametrize("profiled", [False, True])
def test_process_exits_cleanly(hip_lib, tmp_path, profiled):
    cmd = [sys.executable, "-c", SCRIPT]
    if profiled:
        prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
        if not os.path.exists(prof):
            pytest.skip("no rocprofv3")
        cmd = [prof, "--kernel-trace", "-d", str(tmp_path / "prof"), "-o", "x", "--"] + cmd
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=180)
    assert "done" in r.stdout, r.stdout + r.stderr[-2000:]
    assert r.returncode == 0, f"exit code {r.returncode}\n{r.stderr[-2000:]}"
