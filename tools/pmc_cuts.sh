#!/bin/bash
# GPU box: SQ wait/busy counters of the truncated builds -> gpurun_out/pmc_cuts.txt
for k in 0 1 2 3 4 5 6 7; do
  WBC_HIP_LIB=$PWD/build_variants/hcut$k.so python3 tools/pmc.py --out gpurun_out/pmc_cut$k.json --groups "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" -- --variant hex --steps 20 --warmup 3 > /dev/null 2>&1
  python3 -c "
import json; d=json.load(open('gpurun_out/pmc_cut$k.json'))['counters']; g=lambda k: d[k]['mean_per_launch']/1024
print('cut $k  wave_cyc %7.0f wait_any %7.0f wait_inst %6.0f active_valu %7.0f insts_valu %6.0f lds %4.0f salu %5.0f' % (4*g('SQ_WAVE_CYCLES'),4*g('SQ_WAIT_ANY'),4*g('SQ_WAIT_INST_ANY'),4*g('SQ_ACTIVE_INST_VALU'),g('SQ_INSTS_VALU'),g('SQ_INSTS_LDS'),g('SQ_INSTS_SALU')))"
done
