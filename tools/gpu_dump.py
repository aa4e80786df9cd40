#!/usr/bin/env python3
"""(Needs the bring-up build of the library: BRX_BRINGUP=1 python brotli-rs_amd/build.py --force.)
Bring-up: collect parked decoder states (whole-LDS dumps at command boundaries) for tools/asm_emu.py.
Run on the GPU box:  python tools/gpu_dump.py <interval> <max> <out.bin> <stream files...>
Each stream is decoded alone (batch of one) with the C++ command loop re-entered after every command
(BRX_DEBUG_STOP=9); every interval-th parked state of an assembly-eligible meta-block is appended to out.bin."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
interval, mx, out = sys.argv[1], sys.argv[2], sys.argv[3]
os.environ["BRX_DEBUG_STOP"] = "9"
os.environ["BRX_DEBUG_DUMP"] = "%s:%s:%s" % (interval, mx, out)
from brotli_rs_amd import brx  # noqa: E402

if os.path.exists(out):
    os.remove(out)
ctx = brx.Context(0)
for f in sys.argv[4:]:
    data = open(f, "rb").read()
    st, o = ctx.decode(data)
    print(f, "status", st, "out", len(o))
ctx.close()
print("dump bytes", os.path.getsize(out) if os.path.exists(out) else 0)
