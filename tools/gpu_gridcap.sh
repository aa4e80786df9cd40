#!/bin/bash
# A/B: fewer resident waves than streams (BRX_GRID_CAP) for the short-stream classes: do rounds that overlap one
# stream's header with another's fill beat one lock-stepped round?
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in ${WLS:-backward65536x4096 quickfox_repeatedx8192 monkeyx16384}; do
  for cap in ${CAPS:-0 3584 3072 2560 2048 1536 1024}; do
    r=$(BRX_GRID_CAP=$cap timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-copy-path 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_median'], d['bit_exact'])")
    echo "$wl cap=$cap kernel_ms avg/median/bit_exact: $r"
  done
done | tee gpurun_out/${TAG:-r03}_gridcap.txt
