#!/bin/bash
# diagnostic: truncated hex-kernel builds  ->  build_variants/hcut<k>.so   (tools/build_cuts.sh 1 2 3 ...)
cd /root/repo; mkdir -p build_variants
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -shared -DWBC_HCUT=$k -o build_variants/hcut$k.so quadruped_drake_amd/csrc/wbc_kernels.hip quadruped_drake_amd/csrc/wbc_traj.hip 2>/dev/null &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
ls build_variants/hcut*.so
