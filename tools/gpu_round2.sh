#!/bin/bash
# GPU round with the copy-heavy workloads next to the headline one.
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r01}
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
for wl in alice29x4096 backward65536x4096 quickfox_repeatedx8192 compressed_repeatedx4096; do
  echo "== bench $wl"
  extra="--no-cpu-baseline --no-traffic"; [ $wl = alice29x4096 ] && extra=""
  timeout 600 python bench.py --steps ${STEPS:-10} --warmup 2 --workload $wl $extra 2>&1 | tail -1 | tee gpurun_out/bench_${TAG}_$wl.json
done
echo "== bench config5"
timeout 600 python bench.py --steps 5 --warmup 1 --workload config5_1MiBx1024 --no-cpu-baseline --no-traffic 2>&1 | tail -1 | tee gpurun_out/bench_${TAG}_config5_1MiBx1024.json
