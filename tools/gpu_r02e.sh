#!/bin/bash
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r02e}
G=tests/golden/data
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/pytest_${TAG}.txt
for wl in alice29x4096 config5_1MiBx1024 compressed_repeatedx4096; do
  echo "== bench $wl"; timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_${TAG}_${wl}.json | cut -c1-200
done
echo "== dumps (mixed context modes: compressed_repeated)"
timeout 300 python tools/gpu_dump.py 1 40 gpurun_out/dump_cr.bin $G/compressed_repeated.compressed 2>&1 | tail -2
echo "== pc sampling"
g++ -O1 -std=c++17 tools/diag_main.cpp -o tools/diag_main -Lbrotli-rs_amd -lbrx -Wl,-rpath,$PWD/brotli-rs_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
N=1 METHOD=host_trap UNIT=time INTERVAL=1 timeout 300 bash tools/gpu_pcsamp.sh 2>&1 | tail -12
cp gpurun_out/pcsamp.txt gpurun_out/pcsamp_n1.txt 2>/dev/null
N=4096 timeout 300 bash tools/gpu_pcsamp.sh 2>&1 | tail -5
cp gpurun_out/pcsamp.txt gpurun_out/pcsamp_n4096.txt 2>/dev/null
ls -la gpurun_out/pcsamp* 2>/dev/null
