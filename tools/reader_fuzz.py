#!/usr/bin/env python3
"""Differential fuzz of the bounded / pulled reader (brx_stream_new_reader behind brx.Decompressor(streaming=True)) -- the paths round 6
changed: a command taken back because it found no room or no resident input (ring and flush cursor come back), the pause in front of
a meta-block that does not fit behind the window, format errors that read as UnexpectedEOF, truncated sources.  Every round: one
stream of 1 .. 40 MiB made on the GPU (adaptive generator, random meta-block size up to 16 MiB, block switches) or hand-assembled
(tests/craft.py takeback_stream: two literal trees, long inserts, long copies) or of uncompressed meta-blocks, pulled through a source
that returns random-sized pieces, with a random input window (1 .. 8 MiB) and a random command loop (0 = assembly, 6 = C++ only),
read in random-sized pieces; sometimes the source is cut or corrupted, and the reader must deliver the oracle's prefix and status.
Usage: reader_fuzz.py [rounds] [seed]"""
import io
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import craft  # noqa: E402
import oracle_py  # noqa: E402
from brotli_rs_amd import brx  # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "data")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
corpus = b"".join(open(os.path.join(G, f), "rb").read() for f in ("alice29.txt", "lcet10.txt", "plrabn12.txt", "asyoulik.txt"))


class Pieces(io.RawIOBase):
    """A source that hands out what it has in pieces of random size (sometimes one byte, sometimes everything asked for)."""

    def __init__(self, data, r):
        self.data, self.at, self.r = data, 0, r

    def read(self, k=-1):
        if k < 0:
            k = len(self.data)
        mode = self.r.randrange(4)
        k = min(k, 1 if mode == 0 else self.r.randrange(1, 70000) if mode == 1 else k)
        out = self.data[self.at:self.at + k]
        self.at += len(out)
        return out


bad = total = 0
gen_ctx = brx.Context(0)
for r in range(rounds):
    kind = rng.randrange(4)
    if kind == 0:  # text with edits, made on the GPU: meta-blocks of 64 KiB .. 16 MiB
        n = rng.randrange(1 << 20, 40 << 20)
        o = rng.randrange(len(corpus))
        src = bytearray(((corpus[o:] + corpus) * (n // len(corpus) + 2))[:n])
        for _ in range(rng.randrange(0, 2000)):
            src[rng.randrange(n)] = rng.randrange(256)
        src = bytes(src)
        mb = rng.choice((1 << 16, 1 << 20, 3 << 20, 1 << 24))
        comp = gen_ctx.generate_batch([src], metablock_bytes=mb, adaptive=True)[0]
        what = "generated %d B, meta-blocks of %d" % (n, mb)
    elif kind == 1:  # long inserts (the ring is overwritten) + copies of several MiB (no room behind the window)
        lits = rng.choice((3000, 5000, 40000))
        copy = rng.choice((70000, (2 << 20) + 3, (7 << 20) + 5))
        dist = rng.choice((1, 8, 777, 2047, 2048, 2049, 3000))
        comp, src = craft.takeback_stream(rng.randrange(1 << 30), rng.randrange(2, 5), [(lits, copy, min(dist, lits)), (6, 2, min(1500, lits))],
                                          mode=rng.randrange(4))
        what = "takeback %d literals + copy %d from %d" % (lits, copy, dist)
    elif kind == 2:  # inserts of hundreds of KB of input (longer than the reader's margin), ratio 4
        n = rng.choice((300000, 600000, 900000))
        comp, src = craft.takeback_stream(rng.randrange(1 << 30), rng.randrange(6, 14), [(n, rng.choice((4, 300, 5000)), 3000), (6, 2, 1500)], mode=0,
                                          tree_syms=rng.choice((2, 4)))
        what = "takeback inserts of %d literals" % n
    else:  # uncompressed meta-blocks of up to 7 MiB between small compressed ones
        b = craft.Bits()
        craft.stream_header(b, 24)
        src = bytearray()
        for _ in range(rng.randrange(2, 8)):
            data = rng.randbytes(rng.choice((100, 65536, 1 << 20, (7 << 20) + 3)))
            nib = 4 if len(data) <= 1 << 16 else 5 if len(data) <= 1 << 20 else 6
            b.put(0, 1); b.put(nib - 4, 2); b.put(len(data) - 1, 4 * nib); b.put(1, 1)
            b.put(0, (-b.n) % 8)
            b.put_bytes(data)
            src += data
            lit = bytes(rng.randrange(256) for _ in range(40))
            craft.MetaBlock([(lit, 20, 7), (b"xy", 300, 40)], mlen=40 + 20 + 2 + 300).emit(b, False, 0)
            src += lit
            src += (bytes(src[-7:]) * 4)[:20]
            src += b"xy"
            src += (bytes(src[-40:]) * 9)[:300]
        b.put(1, 1); b.put(1, 1)
        comp, src = b.bytes(), bytes(src)
        what = "raw + small meta-blocks, %d B" % len(src)
    # damage
    dmg = rng.randrange(5)
    if dmg == 0:
        comp = comp[:rng.randrange(1, len(comp))]
        what += ", cut"
    elif dmg == 1 and len(comp) > 100:
        c = bytearray(comp)
        for _ in range(rng.randrange(1, 4)):
            c[rng.randrange(len(c) // 2, len(c))] ^= 1 << rng.randrange(8)
        comp = bytes(c)
        what += ", bit flips"
    elif dmg == 2:
        comp = comp + rng.randbytes(rng.randrange(1, 50))
        what += ", trailing bytes"
    want_st, want = oracle_py.decode(comp, 0, cap=len(src) + (1 << 20))[:2]
    window = rng.choice((1 << 20, 2 << 20, 8 << 20))
    loop = rng.choice((0, 0, 6))
    ctx = brx.Context(0, options={"command_loop": loop, "reader_window": window, "reader_mb_room": rng.choice((1, 1, 0))})
    d = brx.Decompressor(Pieces(comp, random.Random(rng.randrange(1 << 30))), ctx, streaming=True)
    got, st = bytearray(), 0
    try:
        while True:
            chunk = d.read(rng.choice((1, 4096, 65536, 1 << 20, 5 << 20)) if len(got) > 64 else 1 << 16)
            if not chunk:
                break
            got += chunk
    except ValueError as e:
        st = [k for k in range(1, 28) if brx.status_str(k) == str(e)][0]
    except brx.BrxError as e:
        st = -1
        what += " LIBRARY ERROR %s" % e
    d.close()
    ctx.close()
    total += 1
    m = min(len(got), len(want))
    ok = st == want_st and (bytes(got) == want if st == 0 else bytes(got[:m]) == want[:m])
    if not ok:
        bad += 1
        print("MISMATCH round %d (%s): status %d want %d, %d bytes want %d, window %d loop %d" % (r, what, st, want_st, len(got), len(want), window, loop))
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "readerfuzz_fail_%d_%d.compressed" % (seed, r)), "wb").write(comp)
    else:
        print("ok round %d (%s): status %d, %d bytes, window %d loop %d" % (r, what, st, len(got), window, loop))
gen_ctx.close()
print("reader_fuzz seed %d: %d streams, %d mismatches" % (seed, total, bad))
sys.exit(1 if bad else 0)
