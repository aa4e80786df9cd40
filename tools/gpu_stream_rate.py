#!/usr/bin/env python3
"""MB/s of ONE stream through the Read facade over a pulled reader (brx_stream_new_reader: bounded memory), for streams whose
meta-blocks are 1 MiB and 16 MiB long (what libbrotlienc makes of a big file at high quality).  Round 6: a slice pauses in front of
a meta-block that does not fit behind the output window and the host makes the room, so the assembly loop runs it; before, such
meta-blocks ran in the C++ loop, one command per call.  usage: gpu_stream_rate.py [MiB]"""
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from brotli_rs_amd import brx  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "data")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 48
corpus = b"".join(open(os.path.join(GOLD, t), "rb").read() for t in ("lcet10.txt", "alice29.txt", "plrabn12.txt", "asyoulik.txt"))
src = bytearray()
k = 0
while len(src) < (mib << 20):
    src += corpus[(k * 18211) % len(corpus):] + corpus[:(k * 18211) % len(corpus)]
    k += 1
src = bytes(src[:mib << 20])
ctx = brx.Context(0)
for room, mb in ((1, 1 << 20), (0, 1 << 24), (1, 1 << 24)):
    ctx.set_option("reader_mb_room", room)
    comp = ctx.generate_batch([src], metablock_bytes=mb, adaptive=True)[0]
    for rep in range(2):
        d = brx.Decompressor(io.BytesIO(comp), ctx, streaming=True)
        t0 = time.perf_counter()
        n, ok, at = 0, True, 0
        while True:
            chunk = d.read(1 << 22)
            if not chunk:
                break
            ok = ok and chunk == src[at:at + len(chunk)]
            at += len(chunk)
        dt = time.perf_counter() - t0
        d.close()
    print("room for whole meta-blocks %d, meta-blocks of %8d B: %d MiB in %.3f s = %.1f MB/s, bit-exact %s, pauses in front of an item that needed room: %d" %
          (room, mb, mib, dt, at / dt / 1e6, ok and at == len(src), ctx.stream_regrown()))
ctx.close()
