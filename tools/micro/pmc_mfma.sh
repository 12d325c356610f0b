#!/bin/bash
# GPU box: PMC counters of the MFMA microbenchmark (two passes) -> gpurun_out/r02b/mfma_micro_pmc.txt
root=$PWD; out=$root/gpurun_out/r02b; mkdir -p $out; rm -f $out/mfma_micro_pmc.txt
$root/build_variants/mfma_micro 1024 2000 > $out/mfma_micro.txt
cd /tmp && export TMPDIR=/tmp
for g in "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_FMA_F64 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  rocprofv3 --kernel-trace --pmc $g -d /tmp/mm -o run --output-format csv -- $root/build_variants/mfma_micro 1024 2000 > /dev/null 2>&1
  python3 - >> $out/mfma_micro_pmc.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/mm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "bench_kernel" in n:
            mode = "mfma" if ("<1>" in n or "ILi1E" in n) else "dpp"
            acc[(mode, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc): print(k[0], k[1], "%.0f" % max(acc[k]), "launches", len(acc[k]))
PY
  rm -rf /tmp/mm
done
cat $out/mfma_micro_pmc.txt
