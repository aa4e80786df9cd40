#!/bin/bash
# kernel time vs streams per launch (occupancy / tail behaviour)
for n in ${NS:-1 256 1024 2048 3072 3840 4096 4352 8192 16384}; do
  timeout 300 python bench.py --workload ${WL:-alice29x4096} --streams $n --steps 5 --warmup 1 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor --verify 0 2>&1 | tail -1 > /tmp/l.json
  python - $n <<'PY'
import sys,json
d=json.load(open('/tmp/l.json')); print(sys.argv[1], d["value"], d["roofline"]["kernel_ms_avg"])
PY
done
