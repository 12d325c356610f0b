#!/bin/bash
# A/B timing of build variants inside one GPU call: tools/r04_ab.sh "<cases>" lib1 lib2 ...   (two rounds, interleaved)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
cases=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    timeout 600 python3 tools/qt.py --lib $lib $cases 2>&1 | grep -v amdgpu.ids
  done
done
