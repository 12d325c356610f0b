#!/bin/bash
# AddressSanitizer + UBSan over the kernel math instantiated on the host (all three lane mappings, every law,
# torque box): the GPU pool has no sanitizer, the headers are the same code.  Exits non-zero on any report.
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -std=c++20 -pthread -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -o /tmp/libhost_tick_asan.so tools/host_tick.cpp
cp tools/libhost_tick.so /tmp/libhost_tick_backup.so
trap 'cp /tmp/libhost_tick_backup.so tools/libhost_tick.so; touch tools/libhost_tick.so' EXIT
cp /tmp/libhost_tick_asan.so tools/libhost_tick.so; touch tools/libhost_tick.so
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest tests/test_kernel_math_host.py tests/test_trajectory.py -q -m "not gpu" -x 2>&1 | grep -v "makecontext" | tail -5
