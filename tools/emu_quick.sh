#!/bin/bash
# The four dump sets of tools/gpu_dumps.sh through tools/asm_emu.py in parallel, sampled (CPU only, ~10 min).
# Usage: tools/emu_quick.sh [every_alice every_c5 every_misc every_xf]
set -u
cd "$(dirname "$0")/.."
G=tests/golden/data
rm -rf /tmp/craft && tar xzf gpurun_out/craft_streams.tgz -C /tmp
(python tools/asm_emu.py ${EMU_SRC:+--src $EMU_SRC} --every ${1:-8} gpurun_out/dump_alice.bin $G/alice29.txt.compressed 2>&1 | tail -3 > /tmp/e_alice.txt) &
(python tools/asm_emu.py ${EMU_SRC:+--src $EMU_SRC} --every ${2:-16} gpurun_out/dump_c5.bin tests/golden/config5/c5_0.compressed 2>&1 | tail -3 > /tmp/e_c5.txt) &
(python tools/asm_emu.py ${EMU_SRC:+--src $EMU_SRC} --every ${3:-16} gpurun_out/dump_misc.bin $G/monkey.compressed $G/asyoulik.txt.compressed $G/compressed_repeated.compressed $G/metablock_reset.compressed tests/golden/enc/e01[5-9]*.compressed tests/golden/enc/e05[0-3]*.compressed tests/golden/enc/e07*.compressed 2>&1 | tail -3 > /tmp/e_misc.txt) &
(python tools/asm_emu.py ${EMU_SRC:+--src $EMU_SRC} --every ${4:-8} gpurun_out/dump_xf.bin /tmp/craft/*.compressed 2>&1 | tail -3 > /tmp/e_xf.txt) &
wait
cat /tmp/e_alice.txt /tmp/e_c5.txt /tmp/e_misc.txt /tmp/e_xf.txt
