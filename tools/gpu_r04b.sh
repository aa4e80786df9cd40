#!/bin/bash
# round 4, launch plans: the tests around hand-overs first, then the whole GPU suite, then the workloads that hand streams over
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
T=${TAG:-x}
timeout 900 python -m pytest tests -m gpu -x -q -k "${K:-wide or level or plan or later or fresh or spill}" 2>&1 | tail -25 | tee gpurun_out/r04/pytest_k_$T.log
if [ "${PYTEST:-1}" = "1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r04/pytest_$T.log; fi
WL="${WL:-mixed_allx4096 mixed_textx4096 lcet10x4096 mapsdatazrhx4096 alice29x4096 backward65536x4096 quickfox_repeatedx8192 config5_1MiBx1024}" STEPS=${STEPS:-6} bash tools/gpu_quick.sh 2>&1 | tee gpurun_out/r04/quick_$T.log
