#!/bin/bash
# A/B of the launch plans (brx_api.cpp launch()): A = catch-all behind the regular kernel, B = pre-pass + all levels next to each other
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in ${WLS:-mixed_allx4096 mixed_textx4096 lcet10x4096 mapsdatazrhx4096 alice29x4096 config5_1MiBx1024}; do
  for v in ${PLANS:-BRX_PLAN_A BRX_PLAN_B BRX_PLAN_B0 DEFAULT}; do
    r=$(env $v=1 timeout 300 python bench.py --workload $wl --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['bit_exact'])")
    echo "$wl $v kernel_ms / ms_per_step / bit_exact: $r"
  done
done | tee gpurun_out/${TAG:-r04}_plan_ab.txt
