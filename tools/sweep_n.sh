#!/bin/bash
# batch-size sweep of the hot kernel: bash tools/sweep_n.sh [config] [variant]
cfg=${1:-3}; var=${2:-auto}
for n in 256 1024 4096 8192 16384 32768 65536; do python bench.py --config $cfg --variant $var --per-gpu $n --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, '%.1f us' % (d['roofline']['kernel_ms']*1e3), '%.1f Mticks/s' % (d['value']/1e6))"; done
