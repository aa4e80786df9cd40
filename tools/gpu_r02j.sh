#!/bin/bash
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r02j}
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/pytest_${TAG}.txt
for wl in farcopy_1MiBx4096 backward65536x4096 quickfox_repeatedx8192 compressed_repeatedx4096; do
  echo "== bench $wl"; timeout 600 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_${TAG}_${wl}.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d['bit_exact'], {k:r.get(k) for k in ('achieved','frac','achieved_physical','frac_physical','kernel_ms_avg')})"
done
