#!/bin/bash
# round-2 assembly loop on hardware: parity suite, benches of every workload, occupancy sweep
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r02d}
echo "== smoke"; timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/pytest_${TAG}.txt
for wl in alice29x4096 config5_1MiBx1024 compressed_repeatedx4096 backward65536x4096 quickfox_repeatedx8192; do
  echo "== bench $wl"; timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_${TAG}_${wl}.json | cut -c1-420
done
echo "== sweep"; NS="1 1024 4096 4352 8192" bash tools/gpu_sweep.sh 2>&1 | tee gpurun_out/sweep_${TAG}.txt
