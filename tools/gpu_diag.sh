#!/bin/bash
BRX_BRINGUP=1 python brotli-rs_amd/build.py --force > /dev/null 2>&1  # statistics / LDS dumps are compiled out of the shipped library
set -u
G=tests/golden/data
for f in monkey.compressed alice29.txt.compressed metablock_reset.compressed backward65536.compressed; do
  echo "== $f"; BRX_DEBUG_STATS=1 timeout 30 ./tools/diag_main $G/$f 1000000 1 2>&1 | tail -3
done
echo "== alice x4096"; BRX_DEBUG_STATS=1 timeout 60 ./tools/diag_main $G/alice29.txt.compressed 152096 4096 2>&1 | tail -4
echo "== pytest"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
