"""GPU: the HIP path against the committed golden fixtures (tests/golden/),
and the raw-vector page loader against pages written by the oracle's
byte-exact page writer."""
import json
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("scan_path")]  # small shapes: exact sums AND the f32 pre-filter (conftest)
GOLD = os.path.join(os.path.dirname(__file__), "golden")
L2, IP, COS = 0, 1, 2


def _same(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and \
        np.array_equal(a[~np.isnan(a)].view(np.uint64), b[~np.isnan(b)].view(np.uint64))


def test_random_small_fixture_on_gpu(hip_lib, oracle_mod):
    from tostore_amd import HipVectorIndex

    z = np.load(os.path.join(GOLD, "random_small.npz"))
    with open(os.path.join(GOLD, "random_small.json")) as f:
        meta = json.load(f)
    rows, queries, keep = z["rows"], z["queries"], z["keep"]
    idx = {m: HipVectorIndex(rows.shape[1], m) for m in (L2, IP, COS)}
    try:
        for m in idx.values():
            m.append(0, rows)
        for m in meta:
            q = queries[m["query"]]
            if m["metric"] == COS:
                q = oracle_mod.normalize_f32(q)
            kp = keep if m["mask"] else None
            ids, dist, cnt = idx[m["metric"]].search(q, m["k"], None, kp)
            assert np.array_equal(ids[0, :cnt[0]], z[m["key"] + "_ids"]), m["key"]
            assert _same(dist[0, :cnt[0]], z[m["key"] + "_dist"]), m["key"]
            ids, dist, cnt = idx[m["metric"]].search(q, m["k"], float.fromhex(m["threshold_hex"]), kp)
            assert np.array_equal(ids[0, :cnt[0]], z[m["key"] + "_thr_ids"]), m["key"]
            assert _same(dist[0, :cnt[0]], z[m["key"] + "_thr_dist"]), m["key"]
    finally:
        for m in idx.values():
            m.close()
    for metric in (L2, IP, COS):
        with HipVectorIndex(rows.shape[1], metric) as bad:
            bad.append(0, z["bad_rows"])
            q = oracle_mod.normalize_f32(queries[0]) if metric == COS else queries[0]
            ids, dist, cnt = bad.search(q, 80)
            assert np.array_equal(ids[0, :cnt[0]], z[f"bad_m{metric}_ids"])
            assert _same(dist[0, :cnt[0]], z[f"bad_m{metric}_dist"])


def test_config_c1_fixture_on_gpu(hip_lib, oracle_mod):
    import hashlib

    from tostore_amd import HipVectorIndex

    with open(os.path.join(GOLD, "config_c1.json")) as f:
        g = json.load(f)
    rows = np.random.Generator(np.random.Philox(20260612)).standard_normal((10000, 128)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True).astype(np.float32)
    qs = np.random.Generator(np.random.Philox(20260613)).standard_normal((4, 128)).astype(np.float32)
    if hashlib.sha256(rows.tobytes()).hexdigest() != g["rows_sha256"]:
        pytest.skip("this NumPy build draws a different Philox stream than the fixture's")
    for metric in (L2, IP, COS):
        with HipVectorIndex(128, metric) as idx:
            idx.append(0, rows)
            for c in [c for c in g["cases"] if c["metric"] == metric]:
                q = oracle_mod.normalize_f32(qs[c["query"]]) if metric == COS else qs[c["query"]]
                ids, dist, cnt = idx.search(q, c["k"])
                assert ids[0].tolist() == c["ids"]
                assert [float(x).hex() for x in dist[0]] == c["dist_hex"]


@pytest.mark.parametrize("precision", [1, 0, 2])
def test_rawvec_partition_file_loader(hip_lib, oracle_mod, tmp_path, precision):
    """<index>/ngh/rawvec/dir_0/p0.ngh as the reference lays it out: page 0 = partition
    meta, data pages 1.. of vectorsPerRawPage slots each (last page partially filled)."""
    from tostore_amd import HipVectorIndex, _ffi

    page_size, dims, n = 16384, 96, 1000
    bpe = {0: 8, 1: 4, 2: 1}[precision]
    vpp = oracle_mod.vectors_per_raw_page(page_size, dims, bpe)
    rng = np.random.default_rng(precision)
    rows = (rng.standard_normal((n, dims)) * 0.3).astype(np.float32)
    n_pages = (n + vpp - 1) // vpp
    path = tmp_path / "p0.ngh"
    expect = np.zeros((n_pages * vpp, dims), np.float32)
    with open(path, "wb") as f:
        f.write(oracle_mod.ngh_meta_page_build(0, 2, n, (n_pages + 1) * page_size, page_size))
        for p in range(n_pages):
            chunk = np.zeros((vpp, dims), np.float32)  # the writer always stores a full-capacity page
            part = rows[p * vpp:(p + 1) * vpp]
            chunk[:len(part)] = part
            page = oracle_mod.rawvec_page_build(chunk, precision, page_size)
            f.write(page)
            expect[p * vpp:(p + 1) * vpp] = oracle_mod.rawvec_page_parse(page, dims, vpp)[0]
    expect = expect[:n]  # what getVectorAsFloat32 yields per node id (i8/f64 pages are lossy / widened)
    q = rng.standard_normal(dims).astype(np.float32)
    with HipVectorIndex(dims, L2) as idx:
        got = idx.load_rawvec_file(str(path), page_size, precision, 0, n)
        assert got == n and idx.size == n
        ids, dist, cnt = idx.search(q, 50)
        eids, edist = oracle_mod.search_exhaustive(expect, q, L2, 50)
        assert np.array_equal(ids[0], eids) and np.array_equal(dist[0], edist)
    # pages past the end of the file: the reference's reader makes up zero vectors there
    # (ngh_partition_manager.dart:276-281); an exhaustive scan must not return those to every query, so
    # their ids stay ABSENT rows (include/tostore_hip.h, tsh_index_load_rawvec_file)
    with HipVectorIndex(dims, L2) as idx:
        got = idx.load_rawvec_file(str(path), page_size, precision, 0, n + 2 * vpp)
        assert got == n_pages * vpp  # the last stored page is full-capacity; nothing beyond it
        assert idx.size == n_pages * vpp
        full = np.zeros((idx.size, dims), np.float32)
        full[:n] = expect
        ids, dist, cnt = idx.search(np.zeros(dims, np.float32), 20)  # the zero query would rank zero rows first
        eids, edist = oracle_mod.search_exhaustive(full, np.zeros(dims, np.float32), L2, 20)
        assert np.array_equal(ids[0], eids) and np.array_equal(dist[0], edist)
        assert ids.max() < n_pages * vpp
    # a page that passes its CRC but is no raw-vector payload (what ciphertext of an encryptVectorIndex
    # database looks like): refused, not loaded as zero rows
    raw = bytearray(open(path, "rb").read())
    off = page_size * 2
    plen = int.from_bytes(raw[off + 8:off + 12], "little")
    raw[off + 20:off + 22] = (60000).to_bytes(2, "little")  # vectorCount the payload cannot hold
    raw[off + 12:off + 16] = oracle_mod.crc32(bytes(raw[off + 20:off + 20 + plen])).to_bytes(4, "little")
    enc = tmp_path / "enc.ngh"
    enc.write_bytes(bytes(raw))
    with HipVectorIndex(dims, L2) as idx:
        with pytest.raises(_ffi.TshError) as e:
            idx.load_rawvec_file(str(enc), page_size, precision, 0, n)
        assert e.value.code == _ffi.TSH_E_FORMAT and "encryptVectorIndex" in e.value.message
        assert idx.size == vpp
    # a page of another type where a raw-vector page belongs
    raw = bytearray(open(path, "rb").read())
    raw[page_size * 2 + 6] = 6  # nghGraph
    other = tmp_path / "other.ngh"
    other.write_bytes(bytes(raw))
    with HipVectorIndex(dims, L2) as idx:
        with pytest.raises(_ffi.TshError) as e:
            idx.load_rawvec_file(str(other), page_size, precision, 0, n)
        assert e.value.code == _ffi.TSH_E_FORMAT
    # a flipped payload byte is a CRC error, as in the reference (btree_page.dart:226-230)
    raw = bytearray(open(path, "rb").read())
    raw[page_size * 2 + 100] ^= 0x40
    bad = tmp_path / "bad.ngh"
    bad.write_bytes(bytes(raw))
    with HipVectorIndex(dims, L2) as idx:
        with pytest.raises(_ffi.TshError) as e:
            idx.load_rawvec_file(str(bad), page_size, precision, 0, n)
        assert e.value.code == _ffi.TSH_E_FORMAT
        assert idx.size == vpp  # page 1 was loaded before the bad page 2


def test_plain_c_driver_without_python_or_torch(hip_lib, tmp_path):
    """tools/cabi_driver.c: a host program with nothing but the C ABI -- the Dart process's situation.  The
    library must bring up HIP on its own, and every stored row must find itself first at distance 0."""
    import os
    import shutil
    import subprocess

    from tostore_amd import build

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = build.build_library()
    exe = tmp_path / "cabi_driver"
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tools", "cabi_driver.c"), "-o", str(exe), "-L", os.path.dirname(so),
                    "-ltostore_hip", "-Wl,-rpath," + os.path.dirname(so), "-lm"], check=True)
    p = subprocess.run([str(exe), "30000", "200", "64", "10"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stdout + p.stderr
    assert "self-hits wrong: 0" in p.stdout and "batches 2" in p.stdout
