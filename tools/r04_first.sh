#!/bin/bash
# round-4 first GPU pass: tests, A/B timing against the round-3 build, stand soak
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04
O=gpurun_out/r04
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt 2>&1
( echo "== new"; timeout 600 python3 tools/qt.py mptc:3:4096 mptc:5:32768 id:2:4096 id:2:32768 pc:3:4096 clf:3:4096 mptc:4:4096 mptc:2:4096 pc:2:4096
  echo "== r03"; timeout 600 python3 tools/qt.py --lib build_variants/libwbc_hip_r03.so mptc:3:4096 mptc:5:32768 id:2:4096 id:2:32768 pc:3:4096 clf:3:4096 mptc:4:4096 mptc:2:4096 pc:2:4096
  echo "== new again"; timeout 600 python3 tools/qt.py mptc:3:4096 id:2:4096 ) > $O/qt_ab.txt 2>&1
( timeout 1500 python3 tools/soak.py 16384 16 3,8,9,12,0,4 ) > $O/soak_first.txt 2>&1
tail -5 $O/pytest_gpu.txt; cat $O/qt_ab.txt; cat $O/soak_first.txt
