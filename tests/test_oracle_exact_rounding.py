"""CPU: every arithmetic step of the oracle is the CORRECTLY ROUNDED IEEE-754 result of the expression the
reference writes.

Dart's double is IEEE-754 binary64 with round-to-nearest-even and no fused multiply-add; a Float32List store rounds
binary64 -> binary32 (RNE).  The reference's loops (/root/reference/lib/src/core/ngh_graph_engine.dart:920-946,
vector_index_manager.dart:1385-1408) are therefore fully determined by those two roundings, and they can be
replayed in EXACT rational arithmetic (fractions.Fraction) with an explicit round-to-nearest-even at every point the
Dart code rounds.  The C oracle (gcc, -ffp-contract=off) must reproduce that replay bit for bit: a contraction
into fma, x87-style excess precision or a reassociated sum would show up here."""
import math
import struct
from fractions import Fraction

import numpy as np
import pytest


def rne(x: Fraction, p: int, emin: int, emax: int) -> float:
    """x rounded to nearest-even in a binary format with p significand bits (p = 53: binary64, 24: binary32),
    smallest normal exponent emin, overflow to inf above emax.  Exact integer arithmetic."""
    if x == 0:
        return 0.0
    sign = -1 if x < 0 else 1
    x = abs(x)
    # exponent e with 2^e <= x < 2^(e+1)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if Fraction(2) ** e > x:
        e -= 1
    elif Fraction(2) ** (e + 1) <= x:
        e += 1
    e_q = max(e, emin) - (p - 1)  # exponent of one unit in the last place (subnormals share emin's)
    q = x / Fraction(2) ** e_q
    n = q.numerator // q.denominator
    r = q - n
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and n % 2 == 1):
        n += 1
    v = Fraction(n) * Fraction(2) ** e_q
    if v >= Fraction(2) ** (emax + 1):
        return sign * math.inf
    return sign * float(v)  # exact: v is representable in binary64


def rn64(x):
    return rne(x, 53, -1022, 1023)


def rn32(x):
    return rne(x, 24, -126, 127)


def sqrt_rn64(x: float) -> float:
    """correctly rounded sqrt of a non-negative finite double, checked in exact arithmetic"""
    r = math.sqrt(x)
    if r == 0.0 or math.isinf(r):
        return r
    X = Fraction(x)
    lo, hi = np.nextafter(r, -np.inf), np.nextafter(r, np.inf)
    # r is correct iff x lies between the midpoints to its neighbours
    assert ((Fraction(lo) + Fraction(r)) / 2) ** 2 <= X <= ((Fraction(hi) + Fraction(r)) / 2) ** 2, "libm sqrt not correctly rounded"
    return r


def replay_sums(q, row, metric):
    """ngh_graph_engine.dart:920-946 in exact arithmetic with the roundings Dart performs"""
    s0 = s1 = mag_a = 0.0
    for a, b in zip(q.tolist(), row.tolist()):
        A, B = Fraction(a), Fraction(b)  # a[i], b[i] widen to double exactly
        if metric == 0:
            diff = rn64(A - B)                        # final diff = a[i] - b[i];
            s0 = rn64(Fraction(s0) + Fraction(rn64(Fraction(diff) * Fraction(diff))))  # sum += diff * diff;
        else:
            s0 = rn64(Fraction(s0) + Fraction(rn64(A * B)))                              # dot += a[i] * b[i];
            if metric == 2:
                mag_a = rn64(Fraction(mag_a) + Fraction(rn64(A * A)))
                s1 = rn64(Fraction(s1) + Fraction(rn64(B * B)))
    return s0, s1, mag_a


def replay_distance(q, row, metric):
    s0, s1, mag_a = replay_sums(q, row, metric)
    if metric == 0:
        return sqrt_rn64(s0)
    if metric == 1:
        return -s0
    denom = rn64(Fraction(sqrt_rn64(mag_a)) * Fraction(sqrt_rn64(s1)))
    sim = rn64(Fraction(s0) / Fraction(denom)) if denom > 0 else 0.0
    return rn64(Fraction(1) - Fraction(sim))


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


def vectors(rng, d, kind):
    if kind == "normal":
        return rng.standard_normal(d).astype(np.float32)
    if kind == "cancel":  # near-cancelling terms of mixed magnitude: every rounding of the running sum matters
        v = (rng.standard_normal(d) * 10.0 ** rng.integers(-6, 7, d)).astype(np.float32)
        v[1::2] = -v[::2][: len(v[1::2])] * np.float32(1 + 2.0 ** -20)
        return v
    if kind == "tiny":
        return (rng.standard_normal(d) * 1e-22).astype(np.float32)
    return (rng.standard_normal(d) * 1e15).astype(np.float32)  # "huge"


def test_rne_helper_against_the_hardware():
    rng = np.random.default_rng(0)
    for x in list(rng.standard_normal(200) * 10.0 ** rng.integers(-30, 30, 200)) + [2.0 ** -149, 2.0 ** -150, 3.4028235677973366e38, 3.5e38, 1 + 2.0 ** -24]:
        with np.errstate(over="ignore"):
            hw = float(np.float32(x))
        assert rn32(Fraction(float(x))) == hw
    a, b = 0.1, 0.7
    assert rn64(Fraction(a) + Fraction(b)) == a + b and rn64(Fraction(a) * Fraction(b)) == a * b
    assert rn64(Fraction(5, 2) * Fraction(2) ** -1074) == 2 * 5e-324  # tie -> even, in the subnormal range


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("kind", ["normal", "cancel", "tiny", "huge"])
def test_exact_sums_and_distances_are_correctly_rounded(oracle_mod, metric, kind):
    from oracle import np_oracle as npo

    rng = np.random.default_rng(100 * metric + len(kind))
    for d in (1, 3, 4, 17, 128, 300):
        q = vectors(rng, d, "normal" if kind != "huge" else "huge")
        row = vectors(rng, d, kind)
        s0, s1, _ = replay_sums(q, row, metric)
        o0, o1 = oracle_mod.exact_sums(q, row, metric)
        assert bits(o0) == bits(s0), (metric, kind, d, "first sum")
        if metric == 2:
            assert bits(o1) == bits(s1), (metric, kind, d, "row norm sum")
        want = replay_distance(q, row, metric)
        got_c = oracle_mod.exact_distance(q, row, metric)
        got_np = float(npo.all_distances(q, row[None, :], metric)[0])
        assert bits(got_c) == bits(want) and bits(got_np) == bits(want), (metric, kind, d, got_c, got_np, want)


def test_to_float32_and_normalize_are_correctly_rounded(oracle_mod):
    from oracle import np_oracle as npo

    rng = np.random.default_rng(5)
    for d in (1, 4, 33, 128):
        vals = (rng.standard_normal(d) * 10.0 ** rng.integers(-3, 4, d)).tolist()
        f = oracle_mod.to_float32(vals, d)  # f32[i] = values[i]  (binary64 -> binary32, RNE)
        assert [float(x) for x in f] == [rn32(Fraction(v)) for v in vals]
        assert np.array_equal(f, npo.to_float32(vals, d))
        # _normalizeFloat32: mag = sqrt(sum v*v) in double; inv = 1.0 / mag; result[i] = v[i] * inv, stored as f32
        mag = 0.0
        for x in f.tolist():
            mag = rn64(Fraction(mag) + Fraction(rn64(Fraction(x) * Fraction(x))))
        mag = sqrt_rn64(mag)
        inv = rn64(Fraction(1) / Fraction(mag))
        want = [rn32(Fraction(rn64(Fraction(x) * Fraction(inv)))) for x in f.tolist()]  # double product, THEN the f32 store
        got = oracle_mod.normalize_f32(f)
        assert [float(x) for x in got] == want
        assert np.array_equal(got, npo.normalize_f32(f))
    z = np.zeros(8, np.float32)
    assert np.array_equal(oracle_mod.normalize_f32(z), z)  # `if (mag == 0) return v;`


def test_distance_to_score_division_and_clamp(oracle_mod):
    # L2: 1.0 / (1.0 + distance): two roundings; cosine: (1.0 - distance).clamp(0.0, 1.0) via compareTo
    for d in (0.0, 0.5, 4.155959577530976, 1e-17, 3.0, 1e300):
        assert bits(oracle_mod.distance_to_score(d, 0)) == bits(rn64(Fraction(1) / Fraction(rn64(Fraction(1) + Fraction(d)))))
    for d, want in ((0.25, 0.75), (-0.5, 1.0), (1.5, 0.0), (1.0, 0.0), (8.881784197001252e-16, 0.9999999999999991)):
        assert bits(oracle_mod.distance_to_score(d, 2)) == bits(want)
