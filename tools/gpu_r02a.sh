#!/bin/bash
# round-2 first GPU call: latency microbenchmarks + the whole GPU parity suite on the round-1 kernel
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== ubench"; timeout 300 tools/ubench/lat 2>&1 | tee gpurun_out/ubench_lat.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/pytest_r02a.txt
