#!/bin/bash
# GPU box: SQ wait / busy / branch counters of the truncated builds (tools/build_cuts.sh) -> gpurun_out/<tag>/pmc_cuts.txt  (tools/pmc_cuts.sh [tag])
# cut k = the tick up to and including phase k: 0 prologue, 1 state, 2 leg, 3 G_b + solve, 4 rows, 5 append, 6 J rows, 7 active set (8 = the whole kernel:
# the product build); differences of consecutive rows = what a phase executes.
tag=${1:-r05}; out=gpurun_out/$tag; mkdir -p $out
for k in 0 1 2 3 4 5 6 7 8; do
  lib=$PWD/build_variants/hcut$k.so; [ $k = 8 ] && lib=$PWD/quadruped_drake_amd/libwbc_hip.so
  WBC_HIP_LIB=$lib python3 tools/pmc.py --out $out/pmc_cut$k.json --groups "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_WAVES" -- --variant hex --steps 20 --warmup 3 > /dev/null 2>&1
  python3 -c "
import json; d=json.load(open('$out/pmc_cut$k.json'))['counters']; g=lambda k: d[k]['mean_per_launch']/1024
print('cut $k  wave_cyc %7.0f wait_any %7.0f wait_inst %6.0f active_valu %7.0f insts_valu %6.0f lds %4.0f salu %5.0f branch %4.0f ifetch %5.0f' % (4*g('SQ_WAVE_CYCLES'),4*g('SQ_WAIT_ANY'),4*g('SQ_WAIT_INST_ANY'),4*g('SQ_ACTIVE_INST_VALU'),g('SQ_INSTS_VALU'),g('SQ_INSTS_LDS'),g('SQ_INSTS_SALU'),g('SQ_INSTS_BRANCH'),g('SQ_IFETCH')))"
done | tee $out/pmc_cuts.txt
