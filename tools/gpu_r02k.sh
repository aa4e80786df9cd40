#!/bin/bash
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r02k}
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/pytest_${TAG}.txt
echo "== bench default (with cpu baseline, config1) + gather"; timeout 600 python bench.py --steps 10 --warmup 2 --gather 2>&1 | tail -1 | tee gpurun_out/bench_${TAG}_default.json | cut -c1-1500
