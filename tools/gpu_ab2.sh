#!/bin/bash
# A/B on one GPU box: every brotli-rs_amd/_ab/libbrx_<name>.so in turn (ROUNDS times, interleaved), kernel ms per workload.
# WLS = "workload[:streams] ..." (streams: override of the workload's stream count, e.g. alice29x4096:1 = one stream alone)
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp brotli-rs_amd/libbrx.so /tmp/libbrx_keep.so
for r in $(seq ${ROUNDS:-2}); do
  for so in brotli-rs_amd/_ab/libbrx_*.so; do
    cp $so brotli-rs_amd/libbrx.so
    for spec in ${WLS:-alice29x4096 config5_1MiBx1024}; do
      wl=${spec%%:*}; n=${spec#*:}; [ "$n" = "$spec" ] && n=""
      timeout 300 python bench.py --workload $wl ${n:+--streams $n} --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor 2>&1 | tail -1 > /tmp/l.json
      python - $so $spec <<'PY'
import sys,json
try:
    d=json.load(open('/tmp/l.json')); print(sys.argv[1].split('libbrx_')[1], sys.argv[2], d["roofline"]["kernel_ms_avg"], d["bit_exact"])
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", open('/tmp/l.json').read()[-300:])
PY
    done
  done
done
cp /tmp/libbrx_keep.so brotli-rs_amd/libbrx.so
