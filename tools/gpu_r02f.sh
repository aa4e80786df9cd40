#!/bin/bash
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r02f}

echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/pytest_${TAG}.txt
for wl in alice29x4096 config5_1MiBx1024 compressed_repeatedx4096; do
  echo "== bench $wl"; timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_${TAG}_${wl}.json | cut -c1-200
done
