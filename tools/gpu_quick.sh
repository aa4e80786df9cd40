#!/bin/bash
# quick kernel-time check of the main workloads (no CPU baseline, no copy_path): WL="a b c" STEPS=n
set -u
cd $GRAFT_REPO_ROOT
for w in ${WL:-alice29x4096 backward65536x4096 quickfox_repeatedx8192 config5_1MiBx1024 monkeyx16384}; do
  timeout 300 python bench.py --workload $w --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('%-26s kernel %.4f ms  value %.0f MB/s  ok=%s' % ('$w', r['roofline']['kernel_ms_avg'], r['value'], r['bit_exact']))"
done
