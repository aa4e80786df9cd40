#!/bin/bash
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
G=tests/golden/data
echo "== dumps"
timeout 300 python tools/gpu_dump.py 97 200 gpurun_out/dump_alice.bin $G/alice29.txt.compressed 2>&1 | tail -3
timeout 300 python tools/gpu_dump.py 997 150 gpurun_out/dump_c5.bin tests/golden/config5/c5_0.compressed 2>&1 | tail -3
timeout 300 python tools/gpu_dump.py 31 150 gpurun_out/dump_misc.bin $G/monkey.compressed $G/asyoulik.txt.compressed $G/compressed_repeated.compressed tests/golden/enc/e01[5-9]*.compressed tests/golden/enc/e05[0-3]*.compressed 2>&1 | tail -6
python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import crafted_sets, os
d='gpurun_out/craft'; os.makedirs(d,exist_ok=True)
for name,s,st,e in crafted_sets.transform_streams():
    if name.endswith('long') and name.split('_')[1] in ('len4','len9','len24'):
        open(os.path.join(d,name+'.compressed'),'wb').write(s)
PY
timeout 300 python tools/gpu_dump.py 7 200 gpurun_out/dump_xf.bin gpurun_out/craft/*.compressed 2>&1 | tail -4
rm -rf gpurun_out/craft
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/pytest_r02b.txt
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -2 | tee gpurun_out/bench_r02b.json
