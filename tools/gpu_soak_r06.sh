#!/bin/bash
# Round-6 extended soak on the FINAL kernel (builder-run, on top of the fixed-seed sections of tests/test_gpu_soak.py): the reader
# fuzz (take-backs, room for whole meta-blocks, cut / corrupted sources) with fresh seeds, and round 5's sections with fresh seeds --
# both launch plans, the sparse-launch build forced, the small slab pool, big pieces -- three processes at a time.
# Usage: tools/gpu_soak_r06.sh "seeds" -> gpurun_out/r06_soak.txt
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/${OUT:-r06_soak.txt}
mkdir -p gpurun_out /tmp/soak
: > $O
run() {  # run "<env>" tool args...  -> one summary line
  local env="$1"; shift
  local tag="$(echo "$env $*" | tr ' =/' '___')"
  ( env $env timeout 900 python tools/$1.py "${@:2}" > /tmp/soak/$tag.log 2>&1; rc=$?; echo "== $env $* :: $(grep -i 'mismatch' /tmp/soak/$tag.log | tail -2 | tr '\n' ' ') rc=$rc" >> $O ) &
}
for SEED in ${1:-701 702}; do
  run "X=1" reader_fuzz 8 $SEED; run "X=1" reader_fuzz 8 $((SEED + 50)); run "X=1" reader_fuzz 8 $((SEED + 100)); wait
  for f in "wide_fuzz 2 $SEED late" "wide_fuzz 2 $SEED" "big_fuzz 3 $SEED" "gen_fuzz 6 $SEED" "small_fuzz 3 $SEED" "device_fuzz 3 $SEED"; do
    for env in BRX_PLAN_A=1 BRX_PLAN_B=1 BRX_LOOP_BUILD=1; do run "$env" $f; done
    wait
  done
  run "BRX_PLAN_A=1" wide_fuzz 1 $SEED big; run "BRX_PLAN_B=1" wide_fuzz 1 $SEED big; run "BRX_GRID_CAP=64" wide_fuzz 2 $SEED; wait
  run "X=1" node_fuzz 30 $SEED; run "X=1" prefix_fuzz 30 $SEED; run "X=1" prefix_fuzz 30 $((SEED + 50)); wait
done
echo "MISMATCH lines: $(grep -c 'MISMATCH' $O)   sections: $(grep -c '^==' $O)" >> $O
tail -3 $O
