#!/bin/bash
# round-4 quick GPU check: GPU suite + kernel times of the short-stream workloads and the headline
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
T=${TAG:-x}
if [ "${PYTEST:-1}" = "1" ]; then timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r04/pytest_$T.log; fi
WL="${WL:-backward65536x4096 quickfox_repeatedx8192 monkeyx16384 quickfoxx16384 ukkonooax16384 alice29x4096}" STEPS=${STEPS:-10} bash tools/gpu_quick.sh 2>&1 | tee gpurun_out/r04/quick_$T.log
