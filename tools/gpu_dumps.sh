#!/bin/bash
# Bring-up: C++-only command loops (parity subset) + whole-LDS state dumps for tools/asm_emu.py (run on the GPU box)
BRX_BRINGUP=1 python brotli-rs_amd/build.py --force > /dev/null 2>&1  # statistics / LDS dumps are compiled out of the shipped library
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
G=tests/golden/data
for stop in 8 7; do
  echo "== parity subset, BRX_DEBUG_STOP=$stop"; BRX_DEBUG_STOP=$stop timeout 600 python tests/gpu_subset_check.py 2>&1 | tail -1
done
timeout 300 python tools/gpu_dump.py 97 200 gpurun_out/dump_alice.bin $G/alice29.txt.compressed 2>&1 | tail -1
timeout 300 python tools/gpu_dump.py 997 150 gpurun_out/dump_c5.bin tests/golden/config5/c5_0.compressed 2>&1 | tail -1
timeout 300 python tools/gpu_dump.py 31 400 gpurun_out/dump_misc.bin $G/monkey.compressed $G/asyoulik.txt.compressed $G/compressed_repeated.compressed $G/metablock_reset.compressed tests/golden/enc/e01[5-9]*.compressed tests/golden/enc/e05[0-3]*.compressed tests/golden/enc/e07*.compressed 2>&1 | tail -1
python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import crafted_sets, craft, os
d='gpurun_out/craft'; os.makedirs(d,exist_ok=True)
for name,s,st,e in crafted_sets.transform_streams() + crafted_sets.transform_edge_streams():
    if 'long' in name and st == 0:
        open(os.path.join(d,name+'.compressed'),'wb').write(s)
for mode in range(4):
    s,e = craft.context_mode_stream(mode, 3, 1500)
    open(os.path.join(d,'ctxmode%d.compressed'%mode),'wb').write(s)
PY
timeout 600 python tools/gpu_dump.py 5 1500 gpurun_out/dump_xf.bin gpurun_out/craft/*.compressed 2>&1 | tail -1
tar czf gpurun_out/craft_streams.tgz -C gpurun_out craft && rm -rf gpurun_out/craft
