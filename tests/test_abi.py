"""CPU: libtostore_hip.so loads without a GPU, exports every symbol that
include/tostore_hip.h declares, refuses compute without a device, and its
host-only merge entry point works."""
import ctypes
import math
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L2, IP, COS = 0, 1, 2


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "tostore_hip.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(tsh_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    from tostore_amd import _ffi

    names = _declared_symbols()
    assert len(names) >= 20
    L = ctypes.CDLL(_ffi.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in tostore_hip.h but not exported"
    assert sorted(_ffi.SIGNATURES) == names, "ctypes SIGNATURES out of sync with the header"
    assert _ffi.lib().tsh_abi_version() == _ffi.ABI_VERSION == 5


def test_no_device_is_an_error_not_a_fallback():
    from tostore_amd import _ffi

    L = _ffi.lib()
    if L.tsh_device_count() > 0:
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert L.tsh_index_create(128, 0, 0, 1, ctypes.byref(h)) == _ffi.TSH_E_NO_DEVICE
    assert not h
    assert "no HIP device" in _ffi.last_error()
    with pytest.raises(_ffi.TshError) as e:
        from tostore_amd import HipVectorIndex
        HipVectorIndex(128, 0)
    assert e.value.code == _ffi.TSH_E_NO_DEVICE


def test_comm_entry_points_without_a_device():
    from tostore_amd import _ffi

    L = _ffi.lib()
    if L.tsh_device_count() > 0:
        pytest.skip("a GPU is present")
    buf = ctypes.create_string_buffer(128)
    assert L.tsh_comm_unique_id(buf) in (_ffi.TSH_E_NO_DEVICE, _ffi.TSH_E_RCCL)
    c = ctypes.c_void_p()
    assert L.tsh_comm_create(buf, 1, 0, 0, ctypes.byref(c)) in (_ffi.TSH_E_NO_DEVICE, _ffi.TSH_E_RCCL)
    assert not c and L.tsh_comm_destroy(None) == 0 and L.tsh_comm_world(None) == 0


def test_argument_validation():
    from tostore_amd import _ffi

    L = _ffi.lib()
    h = ctypes.c_void_p()
    assert L.tsh_index_create(0, 0, 0, 1, ctypes.byref(h)) == _ffi.TSH_E_BAD_ARG
    assert L.tsh_index_create(128, 7, 0, 1, ctypes.byref(h)) == _ffi.TSH_E_BAD_ARG
    assert L.tsh_index_create(128, 0, -1, 1, ctypes.byref(h)) == _ffi.TSH_E_BAD_ARG
    assert L.tsh_index_destroy(None) == 0
    assert L.tsh_index_size(None) == 0
    assert L.tsh_index_append(None, 0, 1, None) == _ffi.TSH_E_BAD_ARG
    assert L.tsh_candidate_block_bytes(256) == 64 + 24 * 256
    assert L.tsh_default_block_entries(100) == 256 and L.tsh_default_block_entries(1000) >= 1156


def _block(entries, cands, k, metric, row_base=0):
    """A candidate block exactly as the device writes it (tsh_kernels.hip.h BlockHeader/BlockEntry)."""
    b = bytearray(64 + 24 * entries)
    struct.pack_into("<8IqqI", b, 0, len(cands), entries, 0, 0, 0, 1 if len(cands) > entries else 0, k, metric,
                     row_base, 0, 0)
    for i, (rid, s0, s1) in enumerate(cands[:entries]):
        struct.pack_into("<qdd", b, 64 + 24 * i, rid, s0, s1)
    return bytes(b)


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_merge_candidates_is_the_reference_ordering(oracle_mod, metric):
    from tostore_amd.sharded import merge_candidate_blocks

    rng = np.random.default_rng(3)
    n, d, k, entries = 500, 32, 20, 128
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[40] = rows[41] = rows[7]  # ties -> id order
    q = rng.standard_normal(d).astype(np.float32)
    if metric == COS:
        q = oracle_mod.normalize_f32(q)
    # three "shards", each contributing its exact local top-k (plus a few extras)
    bounds = [(0, 170), (170, 330), (330, 500)]
    blocks = b""
    for lo, hi in bounds:
        ids, _ = oracle_mod.search_exhaustive(rows[lo:hi], q, metric, k + 5)
        cands = [(lo + int(i),) + oracle_mod.exact_sums(q, rows[lo + int(i)], metric) for i in ids]
        blocks += _block(entries, cands, k, metric, lo)
    for thr in (None, float(oracle_mod.search_exhaustive(rows, q, metric, k)[1][k // 2])):
        ids, dist, cnt = merge_candidate_blocks(metric, d, q, k, thr, np.frombuffer(blocks, np.uint8), 3, entries)
        eids, edist = oracle_mod.search_exhaustive(rows, q, metric, k, thr)
        assert cnt[0] == len(eids)
        assert np.array_equal(ids[0, :cnt[0]], eids)
        assert np.array_equal(dist[0, :cnt[0]].view(np.uint64), edist.view(np.uint64))


def test_merge_of_many_queries_runs_on_the_host_pool(oracle_mod):
    """nq >= 64 goes through the library's pool of polling host threads (jobs complete on items, late workers find
    the job closed): repeated calls, with pauses long enough for the workers to park in between, stay correct"""
    import time

    from tostore_amd.sharded import merge_candidate_blocks

    rng = np.random.default_rng(9)
    n, d, k, entries, nq = 300, 8, 5, 64, 200
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    blocks = b""
    expect = []
    for q in qs:
        ids, _ = oracle_mod.search_exhaustive(rows, q, L2, k + 3)
        blocks += _block(entries, [(int(i),) + oracle_mod.exact_sums(q, rows[int(i)], L2) for i in ids], k, L2)
        expect.append(oracle_mod.search_exhaustive(rows, q, L2, k))
    buf = np.frombuffer(blocks, np.uint8)
    for rep in range(30):
        ids, dist, cnt = merge_candidate_blocks(L2, d, qs, k, None, buf, 1, entries)
        for i in range(nq):
            assert cnt[i] == k and np.array_equal(ids[i], expect[i][0])
            assert np.array_equal(dist[i].view(np.uint64), expect[i][1].view(np.uint64))
        if rep % 10 == 9:
            time.sleep(0.01)  # > the workers' polling window: the next call has to wake them


def test_merge_orders_like_double_compare_to(oracle_mod):
    """the finaliser's integer order keys against the restated double.compareTo on special values (IP: distance =
    -sum0, so every double can be produced exactly): -inf < negatives < -0.0 < +0.0 < positives < +inf < NaN, NaNs
    equal among themselves, ties by id"""
    import functools

    from tostore_amd.sharded import merge_candidate_blocks

    rng = np.random.default_rng(21)
    specials = [0.0, -0.0, math.inf, -math.inf, math.nan, -math.nan, 5e-324, -5e-324, 1.0, -1.0, 1.7976931348623157e308,
                -1.7976931348623157e308, 2.2250738585072014e-308]
    for rep in range(20):
        vals = [float(x) for x in rng.standard_normal(40)] + specials + [rng.choice(specials) for _ in range(20)]
        vals += vals[:10]  # exact duplicates -> id order
        rng.shuffle(vals)
        cands = [(int(i), -v, 0.0) for i, v in enumerate(vals)]  # IP: dist = -s0 = v
        blk = _block(128, cands, len(cands), IP)
        ids, dist, cnt = merge_candidate_blocks(IP, 4, np.zeros(4, np.float32), len(cands), None,
                                                np.frombuffer(blk, np.uint8), 1, 128)
        order = sorted(range(len(vals)), key=functools.cmp_to_key(
            lambda a, b: oracle_mod.compare_double(vals[a], vals[b]) or (a - b)))
        assert cnt[0] == len(vals) and ids[0].tolist() == order
        got = dist[0]
        for g, i in zip(got, order):
            assert (math.isnan(g) and math.isnan(vals[i])) or (g == vals[i] and math.copysign(1, g) == math.copysign(1, vals[i]))


def test_merge_sort_of_clustered_distances(oracle_mod):
    """the finaliser's counting-bucket sort on what a query's candidates look like -- distances that share their
    leading bits, exact ties (id order), a few far outliers -- at list lengths either side of its small-list and
    long-run fallbacks"""
    from tostore_amd.sharded import merge_candidate_blocks

    rng = np.random.default_rng(33)
    for n in (23, 24, 25, 100, 127, 128, 300, 511):
        for spread in (1e-3, 1e-9, 0.0, 10.0):
            base = 0.9 + spread * rng.random(n)
            vals = [float(x) for x in base]
            for _ in range(n // 6):  # exact ties
                vals[int(rng.integers(n))] = vals[int(rng.integers(n))]
            if n > 30 and spread == 1e-3:
                vals[3], vals[7], vals[11] = -1e300, 1e300, math.nan
            cands = [(int(i), -v, 0.0) for i, v in zip(rng.permutation(10 * n)[:n], vals)]  # IP: dist = -s0 = v
            blk = _block(512, cands, n, IP)
            ids, dist, cnt = merge_candidate_blocks(IP, 4, np.zeros(4, np.float32), n, None,
                                                    np.frombuffer(blk, np.uint8), 1, 512)
            key = lambda c: (math.isnan(c[1]), -c[1] if not math.isnan(c[1]) else 0.0, c[0])  # noqa: E731
            want = sorted(cands, key=key)
            assert cnt[0] == n and ids[0].tolist() == [c[0] for c in want], (n, spread)
            for g, c in zip(dist[0], want):
                assert (math.isnan(g) and math.isnan(c[1])) or g == -c[1]
            # the cut: the k smallest of the same list
            k = max(1, n // 3)
            ids2, _, cnt2 = merge_candidate_blocks(IP, 4, np.zeros(4, np.float32), k, None,
                                                   np.frombuffer(_block(512, cands, k, IP), np.uint8), 1, 512)
            assert cnt2[0] == k and ids2[0].tolist() == [c[0] for c in want[:k]]


def test_merge_reports_truncated_blocks(oracle_mod):
    from tostore_amd import _ffi
    from tostore_amd.sharded import merge_candidate_blocks

    cands = [(i, float(i), 0.0) for i in range(300)]
    blk = _block(128, cands, 10, L2)
    with pytest.raises(_ffi.TshError) as e:
        merge_candidate_blocks(L2, 4, np.zeros(4, np.float32), 10, None, np.frombuffer(blk, np.uint8), 1, 128)
    assert e.value.code == _ffi.TSH_E_OVERFLOW and e.value.needed_entries >= 300
    ids, dist, cnt = merge_candidate_blocks(L2, 4, np.zeros(4, np.float32), 10, None,
                                            np.frombuffer(_block(320, cands, 10, L2), np.uint8), 1, 320)
    assert ids[0].tolist() == list(range(10)) and dist[0].tolist() == [math.sqrt(i) for i in range(10)]
    # NaN distances sort last, -0.0 before +0.0 (double.compareTo)
    c2 = [(0, math.nan, 0.0), (1, 4.0, 0.0), (2, math.inf, 0.0)]
    ids, dist, cnt = merge_candidate_blocks(L2, 4, np.zeros(4, np.float32), 3, None,
                                            np.frombuffer(_block(8, c2, 3, L2), np.uint8), 1, 8)
    assert ids[0].tolist() == [1, 2, 0]
    c3 = [(5, 0.0, 0.0), (6, -0.0, 0.0)]  # IP: dist = -s0 -> id 5 gives -0.0, id 6 gives +0.0
    ids, dist, cnt = merge_candidate_blocks(IP, 4, np.zeros(4, np.float32), 2, None,
                                            np.frombuffer(_block(8, c3, 2, IP), np.uint8), 1, 8)
    assert ids[0].tolist() == [5, 6]


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/tostore_hip.h compiles as strict C99 and a C program links against the library: the boundary
    really is a C ABI (what dart:ffi binds).  Without a GPU every compute entry fails loudly."""
    import shutil
    import subprocess

    from tostore_amd import _ffi, build

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    so = build.build_library()
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "tostore_hip.h"\n#include <stdio.h>\n'
        "int main(void) {\n  tsh_index *idx = 0; char buf[256];\n"
        "  int n = tsh_device_count();\n  int rc = tsh_index_create(8, 0, 0, 1, &idx);\n"
        "  tsh_last_error(buf, (int32_t)sizeof buf);\n"
        '  printf("%d %d %d %s\\n", tsh_abi_version(), n, rc, buf);\n'
        "  if (rc == 0) tsh_index_destroy(idx);\n  return 0;\n}\n")
    exe = tmp_path / "abi"
    libdir = os.path.dirname(so)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe), "-L", libdir, "-ltostore_hip", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split(None, 3)
    assert out[0] == str(_ffi.ABI_VERSION)
    if int(out[1]) == 0:
        assert int(out[2]) == _ffi.TSH_E_NO_DEVICE
    else:
        assert int(out[2]) == 0


def test_open_ngh_reads_meta_before_it_needs_a_device(tmp_path):
    """Host logic of the cold start that needs no GPU: meta.json is found, parsed and validated first
    (TSH_E_IO / TSH_E_FORMAT); only a well-formed one gets as far as creating the device index."""
    import json

    from tostore_amd import _ffi

    L = _ffi.lib()
    out = ctypes.c_void_p()

    def rc_of(text):
        d = tmp_path / "ngh"
        d.mkdir(exist_ok=True)
        if text is None:
            if (d / "meta.json").exists():
                (d / "meta.json").unlink()
        else:
            (d / "meta.json").write_text(text)
        return L.tsh_index_open_ngh(str(d).encode(), 0, 1, ctypes.byref(out), None)

    assert rc_of(None) == _ffi.TSH_E_IO
    assert rc_of("[1, 2]") == _ffi.TSH_E_FORMAT
    assert rc_of('{"name": "x"') == _ffi.TSH_E_FORMAT                       # truncated
    assert rc_of('{"name": "x", "tableName": "t"}') == _ffi.TSH_E_FORMAT    # no dimensions
    assert rc_of(json.dumps({"dimensions": 70000})) == _ffi.TSH_E_FORMAT
    assert rc_of(json.dumps({"dimensions": 8, "nghPageSize": 16})) == _ffi.TSH_E_FORMAT
    good = {"name": "i", "dimensions": 8, "nested": {"a": [1, {"b": '}\\"]'}]}, "nextNodeId": 0, "precision": "float32"}
    rc = rc_of(json.dumps(good))
    if L.tsh_device_count() == 0:
        assert rc == _ffi.TSH_E_NO_DEVICE
    else:
        assert rc == 0
        L.tsh_index_destroy(out)


def _comm_id_in_a_process(env_extra, hooks):
    import subprocess
    import sys

    code = ("import ctypes, sys; sys.path.insert(0, %r); from tostore_amd import _ffi; L = _ffi.lib(); %s"
            "b = ctypes.create_string_buffer(128); rc = L.tsh_comm_unique_id(b); print(rc, _ffi.last_error() if rc else '')"
            % (ROOT, "_ffi.enable_test_hooks(); " if hooks else ""))
    env = {k: v for k, v in os.environ.items() if k != "TSH_RCCL_LIB"}
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    rc, _, msg = p.stdout.strip().partition(" ")
    return int(rc), msg


def test_environment_alone_cannot_steer_the_library():
    """An embedded database must not dlopen what its environment names, or fail allocations because a variable says
    so: TSH_RCCL_LIB / TSH_TEST_FAIL_ALLOC_OVER / TSH_SHARDS_SHARE_DEVICES are read only in a process that switched
    the test hooks on itself (tsh_index_set_option(NULL, TSH_OPT_TEST_HOOKS, magic)).  (The RCCL library is chosen
    once per process: processes of their own.)"""
    bad = {"TSH_RCCL_LIB": "/nonexistent/librccl_stand_in.so", "TSH_TEST_FAIL_ALLOC_OVER": "1",
           "TSH_SHARDS_SHARE_DEVICES": "1"}
    plain = _comm_id_in_a_process({}, hooks=False)
    assert _comm_id_in_a_process(bad, hooks=False) == plain, "the variables alone changed the library's behaviour"
    assert "TSH_RCCL_LIB" not in plain[1]
    from tostore_amd import _ffi

    L = _ffi.lib()
    assert L.tsh_index_set_option(None, _ffi.TSH_OPT_TEST_HOOKS, 12345) == _ffi.TSH_E_BAD_ARG  # wrong magic
    assert L.tsh_index_set_option(None, _ffi.TSH_OPT_TEST_HOOKS, 0) == _ffi.TSH_OK


def test_rccl_override_that_does_not_load_is_an_error():
    """With the test hooks on, TSH_RCCL_LIB names the library tsh_comm_* loads instead of librccl (tests/fake_rccl is
    the one user): a path that does not load is TSH_E_RCCL with the reason -- never a silent fall-back to the system's
    librccl."""
    rc, msg = _comm_id_in_a_process({"TSH_RCCL_LIB": "/nonexistent/librccl_stand_in.so"}, hooks=True)
    assert rc == -10 and "TSH_RCCL_LIB" in msg and "does not load" in msg, (rc, msg)


def test_release_library_reads_few_environment_variables():
    """at most six getenv call sites in the release objects (probe switches are compiled out,
    test hooks share one site behind the opt-in)."""
    import re

    csrc = os.path.join(ROOT, "tostore_amd", "csrc")
    sites = []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".h", ".hip")):
            continue
        in_probes = 0
        for ln, line in enumerate(open(os.path.join(csrc, name)), 1):
            t = line.strip()
            if t.startswith("#ifdef TSH_PROBES"):
                in_probes += 1
            elif in_probes and t.startswith("#if"):
                in_probes += 1
            elif in_probes and t.startswith("#endif"):
                in_probes -= 1
            code = line.split("//")[0]
            if not in_probes and re.search(r"(?<![_\w])getenv\(", code):
                sites.append("%s:%d" % (name, ln))
    assert len(sites) <= 6, sites


def test_header_constants_and_the_ctypes_mirror_agree():
    """Every `#define TSH_* <integer>` of include/tostore_hip.h that tostore_amd/_ffi.py mirrors by name carries the same value
    there (status codes, metrics, option numbers, the test hooks' magic), and every option number of the header is mirrored."""
    import re

    from tostore_amd import _ffi

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "tostore_hip.h")).read()
    defs = {}
    for name, val in re.findall(r"^#define (TSH_[A-Z0-9_]+) \(?(-?(?:0x[0-9a-fA-F]+|\d+))(?:ll)?\)?", text, re.M):
        defs[name] = int(val, 0)
    assert defs["TSH_ABI_VERSION"] == _ffi.ABI_VERSION
    mirrored = [n for n in defs if hasattr(_ffi, n)]
    assert len(mirrored) >= 20, mirrored
    for n in mirrored:
        assert getattr(_ffi, n) == defs[n], (n, getattr(_ffi, n), defs[n])
    for n in defs:
        if n.startswith("TSH_OPT_") or n.startswith("TSH_E_") or n.startswith("TSH_METRIC_"):
            assert hasattr(_ffi, n) or n.startswith("TSH_METRIC_"), n + " is not mirrored in _ffi.py"
