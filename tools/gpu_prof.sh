#!/bin/bash
# bring-up: build the loop with its in-loop timers (BRX_PROF) on the GPU box and print them for single streams
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
BRX_PROF=1 python brotli-rs_amd/build.py --force > /dev/null 2>&1
g++ -O1 -std=c++17 tools/diag_main.cpp -o tools/diag_main -Lbrotli-rs_amd -lbrx -Wl,-rpath,$PWD/brotli-rs_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64
for f in tests/golden/data/alice29.txt.compressed tests/golden/config5/c5_0.compressed; do
  for n in 1 4096; do
    [ $n = 4096 ] && [ $f != tests/golden/data/alice29.txt.compressed ] && n=1024
    echo "== $f x $n"; BRX_DEBUG_STATS=1 timeout 120 ./tools/diag_main $f 1048592 $n 2>&1 | grep -v amdgpu | tail -4
  done
done
