#!/bin/bash
# One GPU round: smoke, parity tests, a short bench.  Everything under `timeout` so a hung kernel cannot
# hold the box until gpurun's own limit.
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -15
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
echo "== bench"; timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 ${BENCH_ARGS:-} 2>&1 | tail -5 | tee gpurun_out/bench_last.log
