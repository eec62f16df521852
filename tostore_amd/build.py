"""Builds libtostore_hip.so in-tree with hipcc for gfx950 (no JIT cache: the
built .so travels with the repo snapshot to the GPU box).

Three translation units compile side by side -- the scan kernels' 144 instantiations, the batched path's key
kernels, and the host side with everything else -- and are linked into the one library.  `LAST_BUILD` says
what the last call did ("rebuilt: ..." / "up to date"); __graft_entry__.build() prints it."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
OUT = os.path.join(_HERE, "libtostore_hip.so")
_HDR = os.path.join("..", "..", "include", "tostore_hip.h")
_KERN = ["tsh_kernels.hip.h", "tsh_batch.hip.h", "tsh_launch.h", _HDR]
# translation unit -> what it includes
UNITS = {
    "tsh_scan_tu.hip": _KERN,
    "tsh_batch_tu.hip": _KERN + ["tsh_batch_f16.hip.h", "tsh_batch_f16pp.hip.h"],
    "tsh_lib.hip": _KERN + ["tsh_batch_f16.hip.h", "tsh_batch_f16pp.hip.h", "tsh_host_sync.h", "tsh_pq.hip.h",
                            "tsh_host_batch.inl.h", "tsh_host_coldstart.inl.h", "tsh_host_pq.inl.h",
                            "tsh_host_comm.inl.h"],
}
SOURCES = list(UNITS)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
          "-Wno-unused-value", "-Wno-unused-result", "-Wno-unused-function"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]
LAST_BUILD = "not run"


def _obj(unit: str) -> str:
    return os.path.join(OBJ, unit.replace(".hip", ".o"))


def _newer(path: str, than: float) -> bool:
    return os.path.exists(path) and os.path.getmtime(path) > than


def _unit_stale(unit: str) -> bool:
    o = _obj(unit)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(_newer(os.path.join(CSRC, f), t) for f in [unit] + UNITS[unit])


def _stale() -> bool:
    """The library is older than something it is built from (whether or not the objects are around: a snapshot on
    the GPU box carries the .so but may not carry them)."""
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(_newer(os.path.join(CSRC, f), t) for u in UNITS for f in [u] + UNITS[u])


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    global LAST_BUILD
    if not force and not _stale():
        LAST_BUILD = "up to date (libtostore_hip.so is newer than every source it is built from)"
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    todo = [u for u in UNITS if force or extra_flags or _unit_stale(u)]

    def compile_unit(unit):
        cmd = [hipcc] + CFLAGS + list(extra_flags) + ["-c", "-o", _obj(unit), os.path.join(CSRC, unit)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        list(ex.map(compile_unit, todo))
    cmd = [hipcc] + LDFLAGS + ["-o", OUT + ".tmp"] + [_obj(u) for u in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(OUT + ".tmp", OUT)
    LAST_BUILD = "rebuilt: compiled %s, linked %d objects" % (", ".join(todo) if todo else "nothing", len(UNITS))
    return OUT


if __name__ == "__main__":
    import sys

    print(build_library(force="--incremental" not in sys.argv, verbose=True))
    print(LAST_BUILD)
