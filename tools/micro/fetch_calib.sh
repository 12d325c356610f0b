#!/bin/bash
# GPU box: FETCH_SIZE of the calibration kernels -> gpurun_out/r02/fetch_calib.txt
root=$PWD; mkdir -p $root/gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fc -o run --output-format csv -- $root/build_variants/fetch_calib > /dev/null 2>&1
python3 - > $root/gpurun_out/r02/fetch_calib.txt <<'PY'
import csv, glob
for f in glob.glob("/tmp/fc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "read_" in r["Kernel_Name"]:
            kb = float(r["Counter_Value"])
            print("%-20s FETCH_SIZE %12.0f KB = %.3f x the %d KB actually read" % (r["Kernel_Name"].split("(")[0].split()[-1], kb, kb / (512 * 1024), 512 * 1024))
PY
cat $root/gpurun_out/r02/fetch_calib.txt
