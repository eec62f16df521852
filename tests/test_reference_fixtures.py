"""CPU: the route from "parity unpinned" to "pinned" (VERDICT round 1, item 6).

tests/golden/ref_inputs.json holds inputs only.  tools/dart/gen_fixtures.dart -- run by someone WITH a Dart SDK
inside the reference tree -- feeds them to the reference's own _toFloat32 / _normalizeFloat32 / _exactDistance /
_distanceToScore and double.compareTo and writes tests/golden/ref_outputs.json.  When that file is present the
tests below hold both oracle restatements (C and NumPy) to it bit for bit; it cannot be produced in this image
(no SDK), so until then they check that the two restatements agree on exactly the quantities the file will hold,
and the last test is skipped with that reason."""
import json
import math
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUTS = os.path.join(ROOT, "tests", "golden", "ref_inputs.json")
OUTPUTS = os.path.join(ROOT, "tests", "golden", "ref_outputs.json")


def h2d(h):
    return struct.unpack("<d", struct.pack("<Q", int(h, 16)))[0]


def d2h(x):
    return "%016x" % struct.unpack("<Q", struct.pack("<d", float(x)))[0]


def f2h(x):
    return "%08x" % struct.unpack("<I", struct.pack("<f", np.float32(x)))[0]


def canon(h):
    """NaNs compare by class, not payload (a Dart VM and C may produce different quiet-NaN bit patterns)."""
    v = h2d(h)
    return "nan" if v != v else h


def predict(impl, compare_double, inputs):
    """What gen_fixtures.dart computes, from one oracle restatement."""
    out = []
    for c in inputs["cases"]:
        metric, dim, k = c["metric"], c["dim"], c["k"]
        rows = np.stack([impl.to_float32([h2d(h) for h in r], dim) for r in c["rows_f64_bits"]])
        q = impl.to_float32([h2d(h) for h in c["query_f64_bits"]], dim)
        if metric == 2:
            q = impl.normalize_f32(q)
        dist = impl.all_distances(q, rows, metric)
        thr = None if c["threshold_bits"] is None else h2d(c["threshold_bits"])
        kept = [i for i in range(len(rows)) if not (thr is not None and dist[i] > thr)]
        import functools
        kept.sort(key=functools.cmp_to_key(lambda a, b: compare_double(float(dist[a]), float(dist[b])) or (a > b) - (a < b)))
        out.append({"name": c["name"],
                    "rows_f32_bits": [[f2h(v) for v in r] for r in rows],
                    "query_f32_bits": [f2h(v) for v in q],
                    "dist_bits": [d2h(v) for v in dist],
                    "score_bits": [d2h(impl.distance_to_score(float(v), metric)) for v in dist],
                    "top_ids": kept[:k]})
    ops = [h2d(h) for h in inputs["compare_to_operands_bits"]]
    cmp_m = [[compare_double(a, b) for b in ops] for a in ops]
    scores = {str(m): [d2h(impl.distance_to_score(h2d(h), m)) for h in inputs["score_distances_bits"]] for m in (0, 1, 2)}
    return {"cases": out, "compare_to": cmp_m, "scores_by_metric": scores}


def same(a, b, what):
    assert len(a["cases"]) == len(b["cases"])
    for x, y in zip(a["cases"], b["cases"]):
        assert x["name"] == y["name"]
        assert x["rows_f32_bits"] == y["rows_f32_bits"], (what, x["name"], "_toFloat32 of the rows")
        assert x["query_f32_bits"] == y["query_f32_bits"], (what, x["name"], "_toFloat32 / _normalizeFloat32 of the query")
        assert [canon(h) for h in x["dist_bits"]] == [canon(h) for h in y["dist_bits"]], (what, x["name"], "_exactDistance")
        assert [canon(h) for h in x["score_bits"]] == [canon(h) for h in y["score_bits"]], (what, x["name"], "_distanceToScore")
        assert x["top_ids"] == y["top_ids"], (what, x["name"], "threshold / compareTo order / cut")
    assert a["compare_to"] == b["compare_to"], (what, "double.compareTo")
    for m in ("0", "1", "2"):
        assert [canon(h) for h in a["scores_by_metric"][m]] == [canon(h) for h in b["scores_by_metric"][m]], (what, "scores", m)


def test_inputs_file_is_current():
    inputs = json.load(open(INPUTS))
    assert inputs["format"] == 1 and len(inputs["cases"]) >= 20
    names = [c["name"] for c in inputs["cases"]]
    assert len(set(names)) == len(names)
    assert {c["metric"] for c in inputs["cases"]} == {0, 1, 2}
    assert any(len(c["rows_f64_bits"][0]) > c["dim"] for c in inputs["cases"]), "a truncating _toFloat32 case"
    assert any(len(c["query_f64_bits"]) < c["dim"] for c in inputs["cases"]), "a zero-padding _toFloat32 case"


def test_both_restatements_agree_on_what_the_reference_will_be_asked(oracle_mod):
    from oracle import np_oracle as npo

    inputs = json.load(open(INPUTS))
    a = predict(oracle_mod, oracle_mod.compare_double, inputs)
    b = predict(npo, npo.compare_double, inputs)
    same(a, b, "C vs NumPy restatement")
    # the README example of SURVEY.md section 8c, as a sanity anchor of the prediction code itself
    readme = {c["name"]: c for c in a["cases"]}
    assert [h2d(h) for h in readme["readme_m0"]["dist_bits"]] == [4.155959577530976, 9.482193837605186]
    assert [h2d(h) for h in readme["readme_m2"]["score_bits"]] == [0.9999999999999991, 0.991368662454484]
    nan_row = a["compare_to"][6]
    assert nan_row[6] == 0 and all(v == 1 for i, v in enumerate(nan_row) if i != 6), "NaN.compareTo: greatest, equal to itself"
    assert a["compare_to"][1][0] == -1 and a["compare_to"][0][1] == 1, "-0.0 < +0.0"


def test_oracle_matches_the_reference_outputs(oracle_mod):
    if not os.path.exists(OUTPUTS):
        pytest.skip("tests/golden/ref_outputs.json absent: it is produced by tools/dart/gen_fixtures.dart inside the "
                    "reference tree with a Dart SDK (none in this image) -- parity stays UNPINNED until it is committed")
    from oracle import np_oracle as npo

    inputs, ref = json.load(open(INPUTS)), json.load(open(OUTPUTS))
    same(predict(oracle_mod, oracle_mod.compare_double, inputs), ref, "C oracle vs reference")
    same(predict(npo, npo.compare_double, inputs), ref, "NumPy oracle vs reference")
