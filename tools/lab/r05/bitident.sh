#!/bin/bash
# GPU box: outputs (torques, metrics, statuses) of two builds on every law / config, compared bit for bit:  tools/lab/r05/bitident.sh a.so b.so
a=$1; b=$2; rc=0
for c in mptc:3:4096 mptc:5:8192 mptc:4:4096 id:2:4096 id:3:4096 pc:3:4096 clf:3:4096 mptc:2:4096 pc:2:2048 clf:2:2048 id:4:2048; do
  python3 tools/dump_tau.py --lib $a --out /tmp/bi_a.npy $c 2>/dev/null; python3 tools/dump_tau.py --lib $b --out /tmp/bi_b.npy $c 2>/dev/null
  python3 -c "
import numpy as np,sys
x=np.load('/tmp/bi_a.npy'); y=np.load('/tmp/bi_b.npy')
same=np.array_equal(x,y)
print('%-14s %s  (max abs diff %.3e, %d doubles)'%('$c','BIT-IDENTICAL' if same else 'DIFFERENT',np.abs(x-y).max(),x.size)); sys.exit(0 if same else 1)" || rc=1
done
exit $rc
