#!/bin/bash
# A/B: the level-1 kernel next to the regular one (default) against strictly behind it (BRX_NO_OVERLAP=1).
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in ${WLS:-mixed_textx4096 alice29x4096 lcet10x4096 mapsdatazrhx4096 backward65536x4096 quickfox_repeatedx8192 monkeyx16384 config5_1MiBx1024}; do
  for no in 1 0; do
    if [ $no = 1 ]; then export BRX_NO_OVERLAP=1; else unset BRX_NO_OVERLAP; fi
    r=$(timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['bit_exact'])")
    echo "$wl overlap=$((1-no)) kernel_ms / ms_per_step / bit_exact: $r"
  done
done | tee gpurun_out/${TAG:-r03}_overlap_ab.txt
