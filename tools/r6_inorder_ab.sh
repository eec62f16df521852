#!/bin/bash
# round 6: the wide pick on its scan stream (in order behind E1) against the tail queues, per shape -- same box, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
V=$(ls tostore_amd/csrc/_build/v87e5908ec5/libtostore_hip_v*.so | head -1)
for rep in 1 2; do
  echo "== shipped (pick in order up to 6.5 M floats of scan)"; timeout 600 python tools/r6_lone_probe.py --rounds 1 2>&1 | grep "pick"
  echo "== variant (pick always in order: TSH_X_INORDER_MAX=1e8)"; TSH_LIB_PATH=$V timeout 600 python tools/r6_lone_probe.py --rounds 1 2>&1 | grep "pick"
done
