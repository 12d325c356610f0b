#!/bin/bash
# GPU box: the round's evidence in one call -> gpurun_out/$1 (default r02): smoke, bench line, rocprofv3 kernel stats of the
# same command, PMC passes (separate runs, never combined with sys/hip traces), HBM traffic, batch-size sweeps, other configs.
tag=${1:-r02}; out=gpurun_out/$tag; mkdir -p $out; root=$PWD
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
python bench.py > $out/bench.json 2> $out/bench.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $root/$out/stats -o run --output-format csv -- python3 $root/bench.py --no-cpu-baseline --steps 200 > $root/$out/stats.log 2>&1 )
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
python3 tools/pmc.py --out $out/pmc.json --groups "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" -- --steps 50 --warmup 5 > $out/pmc.log 2>&1
python3 tools/pmc.py --out $out/pmc_n32768.json --groups "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES" -- --config 5 --per-gpu 32768 --steps 20 --warmup 3 > $out/pmc_n32768.log 2>&1
bash tools/sweep_n.sh 3 hex > $out/sweep_mptc.txt 2>&1
bash tools/sweep_n.sh 2 hex > $out/sweep_id.txt 2>&1
python3 tools/qt.py --steps 300 mptc:3:4096 mptc:4:4096 mptc:5:4096 pc:3:4096 clf:3:4096 id:3:4096 > $out/other_cfgs.txt 2>&1
python3 tools/tail_exp.py > $out/tail_experiment.txt 2>&1
python3 tools/rollout_bench.py > $out/rollout.txt 2>&1
cat $out/bench.json | head -c 1500; echo; head -4 $out/kernel_stats.csv; tail -12 $out/pmc.log
