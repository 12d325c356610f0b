#!/bin/bash
# GPU box: two builds under the BENCH protocol (1 s clock ramp, 20 warm-up + 200 timed launches, HIP events) and under rocprofv3 --kernel-trace --stats of the same command,
# interleaved, on one box:  tools/lab/r05/ab_bench.sh "<bench args>" a.so b.so   ->  kernel_ms (HIP events) and the profiler's average per build and round
args=$1; shift
root=$PWD
for round in 1 2; do
  for lib in "$@"; do
    ms=$(WBC_HIP_LIB=$root/$lib python3 bench.py --no-cpu-baseline $args 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f us kernel, %.1f M ticks/s, %.1f M kernel-only' % (d['roofline']['kernel_ms']*1e3, d['value']/1e6, d['value_kernel_only']/1e6))")
    d=$(mktemp -d /tmp/abb.XXXX)
    ( cd /tmp && export TMPDIR=/tmp WBC_HIP_LIB=$root/$lib && rocprofv3 --kernel-trace --stats -d $d -o run --output-format csv -- python3 $root/bench.py --no-cpu-baseline $args > /dev/null 2>&1 )
    avg=$(find $d -name "*kernel_stats.csv" | head -1 | xargs grep wbc_hex_kernel | head -1 | python3 -c "import sys,csv; r=next(csv.reader(sys.stdin)); print('rocprofv3 avg %.2f us over %s calls (min %.2f)' % (float(r[3])/1e3, r[1], float(r[5])/1e3))")
    echo "round $round  $lib  [$args]  $ms | $avg"
    rm -rf $d
  done
done
