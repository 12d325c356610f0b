#!/bin/bash
# diagnostic: truncated hex-kernel builds  ->  build_variants/hcut<k>.so   (tools/build_cuts.sh [-k KIND] 1 2 3 ...)
# one law per build (-DWBC_DEV_ONLY=<kind>, default 1 = MPTC): ~15 s each instead of ~60 s
cd /root/repo; mkdir -p build_variants
kind=1; if [ "$1" = "-k" ]; then kind=$2; shift 2; fi
for k in "$@"; do
  /opt/rocm/bin/hipcc $(cat /root/repo/quadruped_drake_amd/csrc/hipcc_flags.txt) -fPIC -shared -DWBC_DEV_ONLY=$kind -DWBC_HCUT=$k -o build_variants/hcut$k.so quadruped_drake_amd/csrc/wbc_kernels.hip quadruped_drake_amd/csrc/wbc_traj.hip 2>/dev/null &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
ls build_variants/hcut*.so
