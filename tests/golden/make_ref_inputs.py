#!/usr/bin/env python
"""Writes tests/golden/ref_inputs.json: the INPUTS (no expected values) that tools/dart/gen_fixtures.dart feeds
to the real reference functions -- VectorIndexManager._toFloat32 / _normalizeFloat32 / _distanceToScore
(/root/reference/lib/src/core/vector_index_manager.dart:1385-1423) and NghGraphEngine._exactDistance
(lib/src/core/ngh_graph_engine.dart:908-946) -- inside the reference tree with a Dart SDK.  What that script
writes back, tests/golden/ref_outputs.json, is what PINS the oracle to the reference
(tests/test_reference_fixtures.py); neither a Dart SDK nor that file exists in this build image.

Every number travels as the hex image of its IEEE-754 bits (Dart has no hex-float parser): doubles as 16 hex
digits, float32 as 8.

    python tests/golden/make_ref_inputs.py
"""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def d2h(x):
    return "%016x" % struct.unpack("<Q", struct.pack("<d", float(x)))[0]


def case(name, metric, dim, rows, query, k, threshold=None):
    return {"name": name, "metric": metric, "dim": dim, "k": k,
            "threshold_bits": None if threshold is None else d2h(threshold),
            "rows_f64_bits": [[d2h(v) for v in r] for r in rows],
            "query_f64_bits": [d2h(v) for v in query]}


def main():
    rng = np.random.Generator(np.random.Philox(20260701))
    cases = []
    d = 128  # the README / example vectors (example/lib/tostore_example.dart:387-413)
    readme_rows = [[i * 0.01 for i in range(d)], [i * 0.02 + 0.5 for i in range(d)]]
    readme_q = [i * 0.015 for i in range(d)]
    for metric in (0, 1, 2):
        cases.append(case("readme_m%d" % metric, metric, d, readme_rows, readme_q, 5))
    for metric in (0, 1, 2):
        for dim, n in ((4, 9), (7, 33), (128, 10), (200, 6)):
            rows = rng.standard_normal((n, dim)) * rng.choice([1e-3, 1.0, 37.5], size=(n, 1))
            rows[1] = rows[0]  # a tie
            q = rng.standard_normal(dim)
            thr = None if dim != 7 else 2.0
            cases.append(case("random_m%d_d%d" % (metric, dim), metric, dim, rows.tolist(), q.tolist(), 10, thr))
    # _toFloat32: truncate / zero-pad, ties-to-even roundings, overflow to infinity, subnormals
    edge_vals = [1.0 + 2.0 ** -24, 1.0 + 3 * 2.0 ** -24, 1.0 - 2.0 ** -25, 3.4028235677973366e38, 3.5e38, 1e-46,
                 2.0 ** -149, 2.0 ** -150, -0.0, 0.1, 1.0 / 3.0, 16777217.0]
    cases.append(case("tofloat32_long_input", 0, 4, [edge_vals, edge_vals[4:]], edge_vals[::-1], 2))
    cases.append(case("tofloat32_short_input", 2, 16, [edge_vals, [0.5, -0.5]], [1.0, 2.0, 3.0], 2))
    # cosine: zero rows / zero query (denom == 0 -> similarity 0), tiny and huge norms
    z = [0.0] * 8
    cases.append(case("cosine_zero_row", 2, 8, [z, [1e-30] * 8, [1e18] * 8, [1, 0, 0, 0, 0, 0, 0, 0]], [0, 3, 4, 0, 0, 0, 0, 0], 4))
    cases.append(case("cosine_zero_query", 2, 8, [[1, 2, 3, 4, 5, 6, 7, 8], z], z, 2))
    # non-finite elements: ordering by double.compareTo (NaN last, -0 < +0)
    nan, inf = float("nan"), float("inf")
    for metric in (0, 1, 2):
        cases.append(case("nonfinite_m%d" % metric, metric, 4,
                          [[1, 2, 3, 4], [nan, 0, 0, 0], [inf, 0, 0, 0], [-inf, 1, 1, 1], [0, 0, 0, 0], [-0.0, -0.0, -0.0, -0.0],
                           [1e20, 1e20, 0, 0]], [1, 1, 1, 1], 7))
    specials = [0.0, -0.0, 1.0, -1.0, inf, -inf, nan, 5e-324, -5e-324, 1.7976931348623157e308]
    out = {"format": 1,
           "note": "inputs only; expected values come from the reference itself (tools/dart/gen_fixtures.dart)",
           "compare_to_operands_bits": [d2h(v) for v in specials],
           "score_distances_bits": [d2h(v) for v in [0.0, -0.0, 0.5, 1.0, 2.0, -3.0, 800.0, -800.0, 1e-17, inf, -inf, nan, 1.0000000000000002]],
           "cases": cases}
    with open(os.path.join(HERE, "ref_inputs.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", len(cases), "cases,", os.path.getsize(os.path.join(HERE, "ref_inputs.json")), "bytes")


if __name__ == "__main__":
    main()
