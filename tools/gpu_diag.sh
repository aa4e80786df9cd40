#!/bin/bash
set -u
G=tests/golden/data
fail=0
for f in empty.compressed x.compressed 64x.compressed 10x10y.compressed x.compressed.03 quickfox.compressed ukkonooa.compressed monkey.compressed quickfox_repeated.compressed backward65536.compressed zeros.compressed alice29.txt.compressed metablock_reset.compressed; do
  echo "== $f"; timeout 30 ./tools/diag_main $G/$f 1000000 1 2>&1 | tail -2
  if [ ${PIPESTATUS[0]} -ne 0 ]; then fail=1; echo "FAILED/TIMEOUT $f"; break; fi
done
if [ $fail -eq 0 ]; then
  echo "== alice x4096"; timeout 60 ./tools/diag_main $G/alice29.txt.compressed 152096 4096 2>&1 | tail -2
  echo "== smoke"; timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -5
  echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30
fi
