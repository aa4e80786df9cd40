#!/bin/bash
# Round-5 extended soak (builder-run, on top of the fixed-seed slice in tests/test_gpu_soak.py): other seeds, both launch plans, the
# sparse-launch build forced (tree cache, speculative end in the SGPR-window loop), the small slab pool.
# Usage: tools/gpu_soak_r05.sh "seeds" -> gpurun_out/r05_soak.txt
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_soak.txt
: > $O
for SEED in ${1:-101 102}; do
  for env in BRX_PLAN_A=1 BRX_PLAN_B=1 BRX_LOOP_BUILD=1; do
    for f in "wide_fuzz 2 $SEED late" "wide_fuzz 2 $SEED" "big_fuzz 3 $SEED" "gen_fuzz 6 $SEED" "small_fuzz 3 $SEED" "device_fuzz 3 $SEED"; do
      set -- $f
      echo "== $env $f" >> $O
      env $env timeout 900 python tools/$1.py $2 $3 ${4:-} 2>&1 | grep -i "mismatch" | tail -3 >> $O
    done
  done
  for env in BRX_PLAN_A=1 BRX_PLAN_B=1; do  # pieces of 1.2 .. 2.5 MiB (more than 64 trees of a kind, the level-4 instance)
    echo "== $env wide_fuzz 1 $SEED big" >> $O
    env $env timeout 900 python tools/wide_fuzz.py 1 $SEED big 2>&1 | grep -i "mismatch" | tail -3 >> $O
  done
  echo "== BRX_GRID_CAP=64 wide_fuzz 2 $SEED" >> $O
  BRX_GRID_CAP=64 timeout 900 python tools/wide_fuzz.py 2 $SEED 2>&1 | grep -i "mismatch" | tail -3 >> $O
done
grep -c "MISMATCH" $O; tail -4 $O
