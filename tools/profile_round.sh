#!/bin/bash
# GPU box: the round's evidence in one call -> gpurun_out/$1 (default r03):
#   smoke, the bench line, rocprofv3 kernel stats of the SAME command, every product kernel under rocprofv3 (all_kernels.md),
#   PMC passes (separate runs; --pmc is never combined with sys/hip traces) for the headline MPTC kernel, for the ID kernel on
#   BASELINE config 2, and the HBM-traffic counters for the three bench shapes, batch-size sweeps, tail experiment, rollout.
# Under rocprofv3 the program itself follows `--` (python3 <script>): no env / bash -c / launcher hop.
tag=${1:-r06}; out=gpurun_out/$tag; mkdir -p $out; root=$PWD
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
python bench.py > $out/bench.json 2> $out/bench.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $root/$out/stats -o run --output-format csv -- python3 $root/bench.py --no-cpu-baseline --steps 200 > $root/$out/stats.log 2>&1 )
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $root/$out/stats_id -o run --output-format csv -- python3 $root/bench.py --no-cpu-baseline --steps 200 --config 2 > $root/$out/stats_id.log 2>&1 )
find $out/stats_id -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/id_kernel_stats.csv
mkdir -p $out/all
# every product kernel, one profiler run per workload
for w in mptc3 id2 pc3 clf3 id3 anymal4 rand5 rand5_32768 tb_mptc3 rollout_mptc rollout_id lookup integrate; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $root/$out/all/$w -o run --output-format csv -- python3 $root/tools/all_kernels.py $w $root/$out/all > $root/$out/all/$w.log 2>&1 )
done
python3 tools/all_kernels.py --report $out/all > $out/all_kernels.md 2> $out/all_kernels.err
# the per-launch traces are tens of MB each (gpurun copies back <= 64 MiB): keep the profiler's summaries only
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete; find $out -name "*domain_stats.csv" -delete
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
G2="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
G3="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
python3 tools/pmc.py --out $out/pmc.json --groups "FETCH_SIZE" "WRITE_SIZE" "$G1" "$G2" "$G3" -- --steps 50 --warmup 5 > $out/pmc.log 2>&1
python3 tools/pmc.py --out $out/pmc_id.json --groups "FETCH_SIZE" "WRITE_SIZE" "$G1" "$G2" -- --config 2 --steps 50 --warmup 5 > $out/pmc_id.log 2>&1
python3 tools/pmc.py --out $out/pmc_cfg5_n4096.json --groups "FETCH_SIZE" "WRITE_SIZE" -- --config 5 --steps 50 --warmup 5 > $out/pmc_cfg5_n4096.log 2>&1
python3 tools/pmc.py --out $out/pmc_n32768.json --groups "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES" -- --config 5 --per-gpu 32768 --steps 20 --warmup 3 > $out/pmc_n32768.log 2>&1
bash tools/sweep_n.sh 3 hex > $out/sweep_mptc.txt 2>&1
bash tools/sweep_n.sh 2 hex > $out/sweep_id.txt 2>&1
python3 tools/tail_exp.py > $out/tail_experiment.txt 2>&1
python3 tools/rollout_bench.py > $out/rollout.txt 2>&1
python3 tools/singular_sweep.py > $out/singular_envelope.md 2> $out/singular.err
# round 5: the driver's own invocation (K = 20) beside the 200-step line; the C++ host on the same batch; where the launch's tail comes from
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_k20.json 2> $out/bench_k20.err
python bench.py --config 5 --per-gpu 4096 --no-cpu-baseline > $out/bench_cfg5.json 2> $out/bench_cfg5.err
python3 -c "from quadruped_drake_amd import workloads as w; w.dump_batch('/tmp/cfg5_4096.bin', w.make_batch(5, n=4096))" && ./examples/wbc_host --batch /tmp/cfg5_4096.bin --gpus 1 --steps 200 --warmup 20 --repeat 5 > $out/wbc_host.json 2> $out/wbc_host.err
if [ -f build_variants/stamps_mptc.so ]; then
  WBC_HIP_LIB=build_variants/stamps_mptc.so python3 tools/stamp_hex.py 4096 --json $out/stamps_mptc.json > $out/stamps_mptc.txt 2>&1
  WBC_HIP_LIB=build_variants/stamps_mptc.so python3 tools/lab/r05/tail_repeat.py 3 24 > $out/tail_repeat_mptc.txt 2>&1
fi
if [ -f build_variants/stamps_gi_mptc.so ]; then WBC_HIP_LIB=build_variants/stamps_gi_mptc.so python3 tools/lab/r05/who_goes_generic.py > $out/who_goes_generic.txt 2>&1; fi
cat $out/bench.json | head -c 1500; echo; head -4 $out/kernel_stats.csv; head -3 $out/id_kernel_stats.csv; tail -12 $out/pmc.log; cat $out/all_kernels.md
