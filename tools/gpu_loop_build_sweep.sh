#!/bin/bash
# kernel ms of both builds of the command loop (BRX_LOOP_BUILD=0: window in VGPRs, 1: in SGPRs) over streams per launch
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
for wl in alice29x4096 config5_1MiBx1024; do
  for n in ${NS:-256 512 1024 1536 2048 3072 4096}; do
    [ $wl = config5_1MiBx1024 ] && [ $n -gt 2048 ] && continue
    line="$wl n=$n"
    for b in 0 1; do
      BRX_LOOP_BUILD=$b timeout 300 python bench.py --workload $wl --streams $n --steps 5 --warmup 1 --no-cpu-baseline --no-traffic --verify 0 2>&1 | tail -1 > /tmp/l.json
      line="$line  build$b $(python -c "import json; print(json.load(open('/tmp/l.json'))['roofline']['kernel_ms_avg'])")"
    done
    echo "$line"
  done
done
