#!/bin/bash
# A/B on one GPU box: every brotli-rs_amd/_ab/libbrx_<name>.so in turn (ROUNDS times, interleaved), kernel ms per workload
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp brotli-rs_amd/libbrx.so /tmp/libbrx_keep.so
for r in $(seq ${ROUNDS:-2}); do
  for so in brotli-rs_amd/_ab/libbrx_*.so; do
    cp $so brotli-rs_amd/libbrx.so
    for wl in ${WLS:-alice29x4096 config5_1MiBx1024}; do
      timeout 300 python bench.py --workload $wl --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-traffic --no-copy-path --no-chain-floor 2>&1 | tail -1 > /tmp/l.json
      python - $so $wl <<'PY'
import sys,json
d=json.load(open('/tmp/l.json')); print(sys.argv[1].split('libbrx_')[1], sys.argv[2], d["roofline"]["kernel_ms_avg"], d["bit_exact"])
PY
    done
  done
done
cp /tmp/libbrx_keep.so brotli-rs_amd/libbrx.so
