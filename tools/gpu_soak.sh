#!/bin/bash
# Differential soak of the final kernels on one GPU box: both builds of the loop forced in turn (BRX_LOOP_BUILD=0 full-chip,
# 1 sparse-launch), three fuzzers each (tools/big_fuzz.py, wide_fuzz.py, gen_fuzz.py: HIP path vs the oracle).
# Usage: tools/gpu_soak.sh [seed]   -> gpurun_out/soak.txt (copied to profiles/ by hand, named per round)
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
SEED=${1:-3}
: > gpurun_out/soak.txt
for b in 0 1; do
  for f in "big_fuzz 6" "wide_fuzz 2" "gen_fuzz 12"; do
    set -- $f
    echo "== BRX_LOOP_BUILD=$b $1 (seed $SEED)" >> gpurun_out/soak.txt
    BRX_LOOP_BUILD=$b timeout 900 python tools/$1.py $2 $SEED 2>&1 | tail -1 >> gpurun_out/soak.txt
  done
done
cat gpurun_out/soak.txt
