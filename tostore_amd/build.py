"""Builds libtostore_hip.so in-tree with hipcc for gfx950 (no JIT cache: the
built .so travels with the repo snapshot to the GPU box).

Three translation units compile side by side -- the scan kernels' 144 instantiations, the batched path's key
kernels, and the host side with everything else -- and are linked into the one library.  `LAST_BUILD` says
what the last call did ("rebuilt: ..." / "up to date"); __graft_entry__.build() prints it.

Builds with extra compile flags (probe builds: -DTSH_PROBES, -DPP_ISSUE=...) NEVER touch the shipped library or
its objects: they get an object directory and an output of their own (csrc/_build/<tag>/libtostore_hip_<tag>.so,
selected at run time with TSH_LIB_PATH).  Every object directory records the flags its objects were compiled
with; objects whose recorded flags differ from the ones asked for are stale whatever their age."""
from __future__ import annotations

import hashlib
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
    "tsh_lib.hip": _KERN + ["tsh_batch_f16.hip.h", "tsh_batch_f16pp.hip.h", "tsh_exact.hip.h", "tsh_mask.hip.h", "tsh_host_sync.h", "tsh_pq.hip.h",
                            "tsh_host_batch.inl.h", "tsh_host_coldstart.inl.h", "tsh_host_pq.inl.h",
                            "tsh_host_comm.inl.h"],
}
SOURCES = list(UNITS)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
          "-Wno-unused-value", "-Wno-unused-result", "-Wno-unused-function"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]
LAST_BUILD = "not run"


def variant_tag(extra_flags) -> str:
    return "v" + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:10]


def variant_paths(tag: str):
    """(object directory, output path) of a flagged build."""
    d = os.path.join(OBJ, tag)
    return d, os.path.join(d, "libtostore_hip_%s.so" % tag)


def _obj(unit: str, objdir: str = OBJ) -> str:
    return os.path.join(objdir, unit.replace(".hip", ".o"))


def _newer(path: str, than: float) -> bool:
    return os.path.exists(path) and os.path.getmtime(path) > than


def _flags_file(objdir: str) -> str:
    return os.path.join(objdir, "flags.txt")


def _flags_match(objdir: str, flags) -> bool:
    try:
        return open(_flags_file(objdir)).read() == " ".join(flags)
    except OSError:
        return False


def _unit_stale(unit: str, objdir: str, flags) -> bool:
    o = _obj(unit, objdir)
    if not os.path.exists(o) or not _flags_match(objdir, flags):
        return True
    t = os.path.getmtime(o)
    return any(_newer(os.path.join(CSRC, f), t) for f in [unit] + UNITS[unit])


def _stale(out: str, objdir: str, flags) -> bool:
    """The library is older than something it is built from (whether or not the objects are around: a snapshot on
    the GPU box carries the .so but may not carry them), or was built with other flags."""
    if not os.path.exists(out):
        return True
    if os.path.exists(_flags_file(objdir)) and not _flags_match(objdir, flags):
        return True
    t = os.path.getmtime(out)
    return any(_newer(os.path.join(CSRC, f), t) for u in UNITS for f in [u] + UNITS[u])


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), tag: str | None = None) -> str:
    """The shipped library (no extra flags) -> tostore_amd/libtostore_hip.so.  With extra_flags: a variant under
    csrc/_build/<tag>/ (tag defaults to a hash of the flags); the shipped library and its objects are left alone."""
    global LAST_BUILD
    extra_flags = list(extra_flags)
    if extra_flags:
        tag = tag or variant_tag(extra_flags)
        objdir, out = variant_paths(tag)
    else:
        objdir, out = OBJ, OUT
    flags = CFLAGS + extra_flags
    if not force and not _stale(out, objdir, flags):
        LAST_BUILD = "up to date (%s is newer than every source it is built from)" % os.path.basename(out)
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(objdir, exist_ok=True)
    todo = [u for u in UNITS if force or _unit_stale(u, objdir, flags)]
    if todo and not _flags_match(objdir, flags):
        try:
            os.remove(_flags_file(objdir))  # (written again once every object below was compiled with the new flags)
        except OSError:
            pass

    def compile_unit(unit):
        cmd = [hipcc] + flags + ["-c", "-o", _obj(unit, objdir), os.path.join(CSRC, unit)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        list(ex.map(compile_unit, todo))
    with open(_flags_file(objdir), "w") as f:
        f.write(" ".join(flags))
    cmd = [hipcc] + LDFLAGS + ["-o", out + ".tmp"] + [_obj(u, objdir) for u in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(out + ".tmp", out)
    LAST_BUILD = "rebuilt %s: compiled %s, linked %d objects" % (os.path.basename(out), ", ".join(todo) if todo else "nothing",
                                                                len(UNITS))
    return out


if __name__ == "__main__":
    import sys

    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    print(build_library(force="--incremental" not in sys.argv, verbose=True, extra_flags=flags))
    print(LAST_BUILD)
