"""Builds libtostore_hip.so in-tree with hipcc for gfx950 (no JIT cache: the
built .so travels with the repo snapshot to the GPU box)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libtostore_hip.so")
SOURCES = ["tsh_lib.hip"]
HEADERS = ["tsh_kernels.hip.h", "tsh_batch.hip.h", "tsh_batch_f16.hip.h", "tsh_fused.hip.h", "tsh_host_sync.h", "tsh_pq.hip.h", "tsh_host_batch.inl.h", "tsh_host_coldstart.inl.h",
           "tsh_host_pq.inl.h", "tsh_host_comm.inl.h", os.path.join("..", "..", "include", "tostore_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value", "-Wno-unused-result", "-ldl"]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", OUT + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
