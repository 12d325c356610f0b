#!/bin/bash
# GPU box: headline bench line, rocprofv3 kernel stats and PMC passes of the same command -> gpurun_out/r01_hex/
out=gpurun_out/r01_hex; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
python bench.py > $out/bench.json 2> $out/bench.err
root=$PWD
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $root/$out/stats -o run --output-format csv -- python3 $root/bench.py --no-cpu-baseline --steps 200 > $root/$out/stats.log 2>&1 )
python3 tools/pmc.py --out $out/pmc.json --groups "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" -- --steps 50 --warmup 5 > $out/pmc.log 2>&1
bash tools/sweep_n.sh 3 hex > $out/sweep_mptc_hex.txt 2>&1
bash tools/sweep_n.sh 3 quad > $out/sweep_mptc_quad.txt 2>&1
bash tools/sweep_n.sh 2 hex > $out/sweep_id_hex.txt 2>&1
bash tools/sweep_n.sh 2 quad > $out/sweep_id_quad.txt 2>&1
for c in 4 5; do python bench.py --config $c --per-gpu 4096 --no-cpu-baseline --steps 100 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$c', d['roofline']['kernel'], d['roofline']['kernel_ms']*1e3, d['value']/1e6)"; done > $out/other_cfgs.txt 2>&1
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
cat $out/bench.json; head -5 $out/kernel_stats.csv; cat $out/pmc.log | tail -25
