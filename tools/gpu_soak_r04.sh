#!/bin/bash
# Round-4 differential soak: both launch plans forced in turn (BRX_PLAN_A / BRX_PLAN_B, tests/brx_knobs.py), both builds of the loop for
# the wide fuzzer, four fuzzers (HIP path vs the oracle).  Usage: tools/gpu_soak_r04.sh [seed] -> gpurun_out/r04_soak.txt
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
SEED=${1:-5}
O=gpurun_out/r04_soak.txt
: > $O
for plan in BRX_PLAN_A BRX_PLAN_B; do
  for f in "wide_fuzz 2" "big_fuzz 4" "gen_fuzz 8" "small_fuzz 4"; do
    set -- $f
    echo "== $plan=1 $1 $2 (seed $SEED)" >> $O
    env $plan=1 timeout 900 python tools/$1.py $2 $SEED 2>&1 | tail -1 >> $O
  done
done
for b in 0 1; do
  echo "== BRX_LOOP_BUILD=$b wide_fuzz 2 (seed $((SEED+1)), default plan)" >> $O
  BRX_LOOP_BUILD=$b timeout 900 python tools/wide_fuzz.py 2 $((SEED+1)) 2>&1 | tail -1 >> $O
done
# a pool of 64 slabs (BRX_GRID_CAP=64: the spill-slab pool follows the largest grid) under corrupted wide streams: a slab that is not
# given back (the round-4 leak behind a failed header) would stall these rounds
echo "== BRX_GRID_CAP=64 wide_fuzz 3 (seed $((SEED+2)), small slab pool)" >> $O
BRX_GRID_CAP=64 timeout 900 python tools/wide_fuzz.py 3 $((SEED+2)) 2>&1 | tail -1 >> $O
cat $O
