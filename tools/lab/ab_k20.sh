#!/bin/bash
# interleaved bench.py runs at the driver's protocol (K = 20) for several libs
for round in 1 2 3; do
  for lib in "$@"; do
    WBC_HIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['region_us']
print('round $round  %-32s value %.2f M  kernel-only %.2f M  kernel %.3f us  region total %.1f us  device %.1f  fixed %.1f  (queue %.1f stats+wait %.1f gather %.1f close %.1f)' % ('$lib', d['value']/1e6, d['value_kernel_only']/1e6, d['roofline']['kernel_ms']*1e3, r['total'], r['device_time_of_the_K_launches'], r['total']-r['device_time_of_the_K_launches'], r['queue_K_launches'], r['statistics_reduce_and_the_one_wait'], r['gather'], r['closing_bracket']))"
  done
done
