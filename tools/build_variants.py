#!/usr/bin/env python3
"""Links probe variants of libtostore_hip.so side by side: the key-kernel translation unit is recompiled with
-DTSH_PROBES and the given -D flags, the shipped library's other objects are reused (enough for switches the key
kernels' own translation unit reads: TSH_F16_GEN, PP_ISSUE, PP_K; probes that need the host side's hooks -- TSH_F16_DBG --
want the whole library as a variant: `python -m tostore_amd.build -DTSH_PROBES`, which never touches the shipped
library either).  Usage:
  python tools/build_variants.py name1:-DPP_ISSUE=1 name2:-DPP_ISSUE=2,-DFOO ...
-> tostore_amd/csrc/_build/var_<name>.so, selected at run time with TSH_LIB_PATH."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tostore_amd import build as B

def one(spec):
    name, _, flags = spec.partition(":")
    flags = [f for f in flags.split(",") if f]
    obj = os.path.join(B.OBJ, f"var_{name}.o")
    out = os.path.join(B.OBJ, f"var_{name}.so")
    subprocess.run(["hipcc"] + B.CFLAGS + ["-DTSH_PROBES"] + flags + ["-c", "-o", obj, os.path.join(B.CSRC, "tsh_batch_tu.hip")], check=True, cwd=B.CSRC)
    objs = [obj if u == "tsh_batch_tu.hip" else B._obj(u) for u in B.UNITS]
    subprocess.run(["hipcc"] + B.LDFLAGS + ["-o", out] + objs, check=True, cwd=B.CSRC)
    return out

with ThreadPoolExecutor(max_workers=4) as ex:
    for o in ex.map(one, sys.argv[1:]):
        print(o)
