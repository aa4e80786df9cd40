#!/bin/bash
# Bring-up: per-phase cycle timers of single streams (needs the BRX_BRINGUP=1 build of the library, made here).
set -u
cd $GRAFT_REPO_ROOT
BRX_BRINGUP=1 python brotli-rs_amd/build.py --force > /dev/null 2>&1
g++ -O2 -std=c++17 tools/diag_main.cpp -o tools/diag_main -Lbrotli-rs_amd -lbrx -Wl,-rpath,$GRAFT_REPO_ROOT/brotli-rs_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
G=tests/golden/data
for f in ${FILES:-backward65536 quickfox_repeated ukkonooa monkey alice29.txt}; do
  for n in 1 ${N2:-4096}; do
    echo "== $f x $n"; BRX_DEBUG_STATS=1 timeout 120 ./tools/diag_main $G/$f.compressed 1048592 $n 3 2>&1 | grep -v amdgpu | grep "phases\|kernel" | tail -3
  done
done
