#!/bin/bash
# a 1 GiB stream (as much compressed input) through host/decompressor.hpp, pulled; resident memory on both sides.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/s1g
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import craft
for tag, data in zip(("prefix", "unit", "final", "unit_out"), craft.periodic_stream_parts(5)):
    open("/tmp/s1g/%s.bin" % tag, "wb").write(data)
PY
g++ -O1 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/stream_reader_test.cpp -o /tmp/s1g/t -Lbrotli-rs_amd -lbrx -Wl,-rpath,$GRAFT_REPO_ROOT/brotli-rs_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64
K=${K:-16320}
echo "# tools/gpu_stream_1gib.sh: prefix + unit x $K + final through brotli::Decompressor<PeriodicReader> (unit: $(stat -c %s /tmp/s1g/unit.bin) B compressed -> $(stat -c %s /tmp/s1g/unit_out.bin) B)" | tee gpurun_out/r04_stream_1GiB.txt
timeout 900 /tmp/s1g/t /tmp/s1g/prefix.bin /tmp/s1g/unit.bin /tmp/s1g/final.bin /tmp/s1g/unit_out.bin $K 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_stream_1GiB.txt
