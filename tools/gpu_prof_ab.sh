#!/bin/bash
# bring-up: in-loop timers (BRX_PROF builds, made on the GPU box) of several brx_hot.S variants
# (brotli-rs_amd/_ab/brx_hot_<name>.S), one stream alone and the full batch
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp brotli-rs_amd/csrc/brx_hot.S /tmp/keep.S; cp brotli-rs_amd/libbrx.so /tmp/keep.so
for s in brotli-rs_amd/_ab/brx_hot_*.S; do
  cp $s brotli-rs_amd/csrc/brx_hot.S
  BRX_PROF=1 python brotli-rs_amd/build.py --force > /dev/null 2>&1
  g++ -O1 -std=c++17 tools/diag_main.cpp -o tools/diag_main -Lbrotli-rs_amd -lbrx -Wl,-rpath,$PWD/brotli-rs_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64
  for n in 1 4096; do
    echo "== $s alice29 x $n"; BRX_DEBUG_STATS=1 timeout 120 ./tools/diag_main tests/golden/data/alice29.txt.compressed 152096 $n 2>&1 | grep -v amdgpu | tail -4
  done
done
cp /tmp/keep.S brotli-rs_amd/csrc/brx_hot.S; cp /tmp/keep.so brotli-rs_amd/libbrx.so
