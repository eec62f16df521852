"""Test-only stand-in for librccl (fake_rccl.c): lets the RCCL branch of tsh_search_sharded run with several ranks on
ONE GPU.  `build()` compiles it in-tree (so it travels to the GPU box with the snapshot) and returns its path;
a process selects it with TSH_RCCL_LIB=<path> before its first tsh_comm_* call.  The product never loads it on its
own."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "fake_rccl.c")
OUT = os.path.join(_HERE, "libfake_rccl.so")


def build() -> str:
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ["gcc", "-O2", "-Wall", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", SRC,
           "-o", OUT + ".tmp", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lrt"]
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT
