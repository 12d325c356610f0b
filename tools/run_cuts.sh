#!/bin/bash
# time the truncated builds (GPU box):  tools/run_cuts.sh "1 2 3 ..." "256 4096" [config]
cfg=${3:-3}
for k in $1; do
  for n in $2; do
    WBC_HIP_LIB=$PWD/build_variants/hcut$k.so python bench.py --config $cfg --variant hex --per-gpu $n --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cut $k n $n  %.1f us' % (d['roofline']['kernel_ms']*1e3))"
  done
done
