#!/bin/bash
# Run tools/asm_emu.py over the LDS dumps collected by tools/gpu_dumps.sh (CPU only).  Usage: tools/emu_all.sh [every]
set -u
cd "$(dirname "$0")/.."
E=${1:-1}
G=tests/golden/data
rm -rf /tmp/craft && tar xzf gpurun_out/craft_streams.tgz -C /tmp
python tools/asm_emu.py --every $E gpurun_out/dump_alice.bin $G/alice29.txt.compressed 2>&1 | tail -3
python tools/asm_emu.py --every $E gpurun_out/dump_c5.bin tests/golden/config5/c5_0.compressed 2>&1 | tail -3
python tools/asm_emu.py --every $E gpurun_out/dump_misc.bin $G/monkey.compressed $G/asyoulik.txt.compressed $G/compressed_repeated.compressed $G/metablock_reset.compressed tests/golden/enc/e01[5-9]*.compressed tests/golden/enc/e05[0-3]*.compressed tests/golden/enc/e07*.compressed 2>&1 | tail -3
python tools/asm_emu.py --every $E gpurun_out/dump_xf.bin /tmp/craft/*.compressed 2>&1 | tail -3
