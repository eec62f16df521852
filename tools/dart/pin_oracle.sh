#!/bin/bash
# One command from "parity unpinned" to "pinned", for anyone with a Dart SDK (>= 3.5.4, the reference's own floor):
#     tools/dart/pin_oracle.sh /path/to/a/checkout/of/tocreator/tostore      (v3.2.0)
# It runs the REFERENCE'S OWN private methods (_toFloat32, _normalizeFloat32, _exactDistance, _distanceToScore, through
# dart:mirrors) and double.compareTo on the inputs of tests/golden/ref_inputs.json, writes what they return, bit for
# bit, to tests/golden/ref_outputs.json, and runs the tests that hold both oracle restatements to that file.
# Commit tests/golden/ref_outputs.json (data only).  What flips: tests/test_reference_fixtures.py::
# test_oracle_matches_the_reference_outputs goes from "skipped (no ref_outputs.json)" to passed, and the "parity unpinned"
# line of oracle/vs_oracle.c, DESIGN.md section 1 and README.md can be struck.
set -e
REF=${1:?usage: tools/dart/pin_oracle.sh <tostore checkout>}
HERE=$(cd "$(dirname "$0")/../.." && pwd)
command -v dart >/dev/null || { echo "no dart on PATH (the build image of this repository has none either)"; exit 2; }
test -f "$REF/lib/src/core/ngh_graph_engine.dart" || { echo "$REF is not a tostore checkout"; exit 2; }
mkdir -p "$REF/tool"
cp "$HERE/tools/dart/gen_fixtures.dart" "$REF/tool/gen_fixtures.dart"
(cd "$REF" && dart pub get && dart run tool/gen_fixtures.dart "$HERE/tests/golden/ref_inputs.json" "$HERE/tests/golden/ref_outputs.json")
rm -f "$REF/tool/gen_fixtures.dart"
cd "$HERE" && python -m pytest tests/test_reference_fixtures.py -q -rs
