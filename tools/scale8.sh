#!/bin/bash
# 8-GPU hand-over kit, part 2: ONE command for a person with a multi-GPU MI355X node.
#
#   tools/scale8.sh [outdir]            (from the repository root; default outdir profiles/scale8)
#   tools/scale8.sh outdir --share-one-gpu     a ONE-GPU box: N = 2, 4, 8 REAL rank processes of bench.py sharing GPU 0 (gloo instead of RCCL: RCCL
#                                              refuses two ranks on one device); sharding, windows, launcher and statistics are the N-GPU run's own
#
# Builds everything, then for N in 1 2 4 8 (as many as the node has) steps the sharded BASELINE configs[4] batch (4096 instances per GPU) through
#   * bench.py --gpus N          one process per GPU (torch.distributed over RCCL only for the end-of-rollout statistics), and
#   * examples/wbc_host --gpus N one C++ process, one thread per GPU, ncclCommInitAll + one ncclAllGather of 22 doubles,
# writes one JSON per run and compares every run with profiles/scale8_expected.json (tools/scale8_expected.py: the per-rank torque checksums and the
# exact statistics, computed beforehand on ONE GPU window by window).  A correct node prints "OK" on every line; the table is the scaling curve
# (ticks/s, per-rank kernel_ms, all-gather microseconds).  Nothing here needs the network or the reference.
set -u
cd "$(dirname "$0")/.."
out=${1:-profiles/scale8}
share=${2:-}
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python3 -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1 || { echo "build failed: $out/build.log"; exit 1; }
ngpu=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $ngpu"
files=()
for n in 1 2 4 8; do
  if [ "$share" = "--share-one-gpu" ] && [ "$n" -gt "$ngpu" ]; then
    python3 bench.py --gpus $n --config 5 --per-gpu 4096 --backend gloo --share-gpu --steps 50 --warmup 5 --no-cpu-baseline > "$out/bench_shared_$n.json" 2> "$out/bench_shared_$n.err" \
      && files+=("$out/bench_shared_$n.json") || echo "bench.py --gpus $n --share-gpu FAILED: $out/bench_shared_$n.err"
    continue
  fi
  [ "$n" -le "$ngpu" ] || { echo "N=$n skipped: only $ngpu GPU(s)"; continue; }
  # bench.py launches its own ranks for N > 1 (python -m torch.distributed.run --nproc-per-node N ... on 127.0.0.1)
  python3 bench.py --gpus $n --config 5 --per-gpu 4096 --steps 200 --warmup 20 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err" \
    && files+=("$out/bench_$n.json") || echo "bench.py --gpus $n FAILED: $out/bench_$n.err"
  python3 -c "
from quadruped_drake_amd import workloads
workloads.dump_batch('$out/cfg5_$n.bin', workloads.make_batch(5, n=4096 * $n))"
  examples/wbc_host --batch "$out/cfg5_$n.bin" --gpus $n --steps 200 --warmup 20 --repeat 5 > "$out/host_$n.json" 2> "$out/host_$n.err" \
    && files+=("$out/host_$n.json") || echo "wbc_host --gpus $n FAILED: $out/host_$n.err"
  rm -f "$out/cfg5_$n.bin"
done
python3 tools/scale8_expected.py --check "${files[@]}" | tee "$out/table.txt"
rc=${PIPESTATUS[0]}
[ "$rc" -eq 0 ] && echo "scale8: every run matches its prediction" || echo "scale8: MISMATCH (see above)"
exit $rc
