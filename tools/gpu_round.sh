#!/bin/bash
# One GPU round: parity tests, bench, rocprofv3 kernel trace (+ optional PMC passes).  Everything under
# `timeout` so a hung kernel cannot hold the box until gpurun's own limit.
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r01}
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== pytest gpu, classification pre-pass + all levels next to each other on every launch (plan B)"; BRX_PLAN_B=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== bench"; timeout 600 python bench.py --steps ${STEPS:-10} --warmup 2 ${BENCH_ARGS:-} 2>&1 | tail -3 | tee gpurun_out/bench_${TAG}.json
if [ "${PROF:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  echo "== rocprofv3 kernel-trace"
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-traffic 2>&1 | tail -3
  if [ -n "${PMC:-}" ]; then
    i=0
    for set in "$PMC" ${PMC2:+"$PMC2"} ${PMC3:+"$PMC3"}; do
      i=$((i+1))
      echo "== rocprofv3 pmc pass $i: $set"
      timeout 600 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --verify 0 2>&1 | tail -2
    done
  fi
  cd $GRAFT_REPO_ROOT
  find gpurun_out/prof_${TAG} gpurun_out/pmc_${TAG}_* -name "*.csv" 2>/dev/null | head -20
  find gpurun_out -name "*_kernel_stats.csv" | head -3 | xargs -r head -5
fi
