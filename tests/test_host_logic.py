"""CPU: the host-side mirror of VectorIndexManager's arithmetic around the seam
(product code in tostore_amd/vector_index_manager.py) against the oracle."""
import math
import os

import numpy as np

L2, IP, COS = 0, 1, 2


def test_to_float32_truncate_pad_round(oracle_mod):
    from tostore_amd import to_float32

    rng = np.random.default_rng(0)
    vals = rng.standard_normal(20) * 1e3
    for dim in (5, 20, 31):
        a, b = to_float32(vals, dim), oracle_mod.to_float32(vals, dim)
        assert a.dtype == np.float32 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert to_float32([0.1], 3).tolist() == [np.float32(0.1), 0.0, 0.0]
    assert to_float32([1e40, -1e40], 2).tolist() == [math.inf, -math.inf]  # f64 -> f32 overflow
    assert to_float32([], 2).tolist() == [0.0, 0.0]


def test_normalize_matches_reference_order(oracle_mod):
    from tostore_amd import normalize_float32

    rng = np.random.default_rng(1)
    for d in (1, 7, 128, 768):
        v = (rng.standard_normal(d) * rng.uniform(1e-3, 1e3)).astype(np.float32)
        a, b = normalize_float32(v), oracle_mod.normalize_f32(v)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    z = np.zeros(8, np.float32)
    assert normalize_float32(z) is z or np.array_equal(normalize_float32(z), z)  # zero vector unchanged


def test_distance_to_score(oracle_mod):
    from tostore_amd import distance_to_score

    xs = [0.0, -0.0, 1e-16, 0.3, 1.0, 1.0 + 1e-12, 2.0, 5.0, -2.5, -103.63200000291876, 700.0, -800.0,
          math.inf, -math.inf, math.nan]
    for metric in (L2, IP, COS):
        for x in xs:
            a, b = distance_to_score(x, metric), oracle_mod.distance_to_score(x, metric)
            assert (math.isnan(a) and math.isnan(b)) or (a == b and math.copysign(1, a) == math.copysign(1, b)), \
                (metric, x, a, b)


def test_flagged_builds_never_touch_the_shipped_library(tmp_path, monkeypatch):
    """tostore_amd/build.py: a build with extra flags gets an object directory and an output of its
    own, and objects compiled with other flags are stale whatever their age."""
    from tostore_amd import build as B

    tag = B.variant_tag(["-DTSH_PROBES"])
    objdir, out = B.variant_paths(tag)
    assert out != B.OUT and os.path.dirname(out) == objdir and objdir.startswith(B.OBJ + os.sep)
    assert B.variant_tag(["-DTSH_PROBES", "-DPP_ISSUE=1"]) != tag
    # the staleness rule, on a scratch object directory
    d = tmp_path / "objs"
    d.mkdir()
    flags = B.CFLAGS + ["-DTSH_PROBES"]
    for u in B.UNITS:
        (d / u.replace(".hip", ".o")).write_bytes(b"x")
        os.utime(d / u.replace(".hip", ".o"), (4e9, 4e9))  # newer than every source
    assert all(B._unit_stale(u, str(d), flags) for u in B.UNITS)      # no flags recorded: stale
    (d / "flags.txt").write_text(" ".join(flags))
    assert not any(B._unit_stale(u, str(d), flags) for u in B.UNITS)  # same flags, newer than the sources: current
    assert all(B._unit_stale(u, str(d), B.CFLAGS) for u in B.UNITS)   # other flags: stale whatever the age
