#!/usr/bin/env python3
"""Soak of the LEAN instance of the kernel (short streams, brx_small.h) on a GPU box: thousands of libbrotlienc-made streams of
a few bytes to a few hundred bytes (text snippets, runs, periodic data, random bytes; qualities 0-11, every window size,
NPOSTFIX / NDIRECT, forced flushes = several meta-blocks), ragged batches with unaligned output slots, compared with the
original bytes; plus corrupted variants (bit flips / truncation) compared with the oracle's status and bytes.  Reports how many
streams the lean instance decoded itself and how many it left to the regular kernel.  Usage: small_fuzz.py [rounds] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import brotli_enc  # noqa: E402
import oracle_py  # noqa: E402
from brotli_rs_amd import brx  # noqa: E402
import brx_knobs  # noqa: E402
import fuzz_slots  # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "data")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
pool = [open(os.path.join(G, f), "rb").read() for f in ("alice29.txt", "lcet10.txt", "plrabn12.txt", "asyoulik.txt")]
assert brotli_enc.available()
ctx = brx_knobs.context(0)
bad = 0
lean_done = lean_left = 0
for r in range(rounds):
    datas, streams = [], []
    for it in range(3000):
        kind = rng.randrange(6)
        if kind == 0:  # a text snippet (dictionary words + transforms at quality >= 5)
            base = rng.choice(pool)
            n = rng.randrange(0, 900)
            o = rng.randrange(len(base) - n)
            data = base[o:o + n]
        elif kind == 1:
            data = bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 300)))
        elif kind == 2:  # a fill: long output from a few bytes (periodic_fill, the tails around it)
            unit = bytes(rng.getrandbits(8) for _ in range(rng.choice([1, 1, 2, 3, 5, 16, 43, 64, 100, 257])))
            n = rng.randrange(1, 400000 if it % 7 == 0 else 9000)
            data = bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 40))) + (unit * (1 + n // len(unit)))[:n]
        elif kind == 3:
            data = bytes(rng.choice(b"ab\n ") for _ in range(rng.randrange(1, 1500)))
        elif kind == 4:  # words in title / upper case: the other transforms
            base = rng.choice(pool)
            o = rng.randrange(len(base) - 600)
            data = base[o:o + rng.randrange(1, 300)].upper() + base[o:o + rng.randrange(1, 300)].title()
        else:  # snippets glued together: back-references at all distances inside a small window
            base = rng.choice(pool)
            data = b"".join(base[o:o + 40] for o in (rng.randrange(len(base) - 40) for _ in range(rng.randrange(1, 12)))) * rng.randrange(1, 4)
        npf = rng.choice([None, 0, 1, 2, 3])
        nd = None if npf is None else rng.randrange(0, 16) << npf
        comp = brotli_enc.compress(data, quality=rng.randrange(0, 12), lgwin=rng.randrange(10, 25), mode=rng.randrange(3),
                                   npostfix=npf, ndirect=nd, flush_every=rng.choice([0, 0, 0, 100, 700]))
        datas.append(data)
        streams.append(comp)
    caps = [len(x) + rng.randrange(0, 40) for x in datas]  # ragged slots -> every 16-byte skew of the output pointer
    outs, status, out_len = ctx.decode_batch(streams, caps)
    left = ctx.last_lean_listed()
    lean_left += left
    lean_done += len(streams) - left
    for i, (d, o, st) in enumerate(zip(datas, outs, status)):
        if st != 0 or o != d:
            bad += 1
            print("MISMATCH valid stream", r, i, int(st), len(d), len(streams[i]), streams[i][:24].hex())
    # corrupted variants: status must equal the oracle's, and so must the bytes of the ones that still decode
    cs = []
    for it in range(3000):
        s = bytearray(rng.choice(streams))
        if not s:
            continue
        if rng.random() < 0.6:
            for _ in range(rng.randrange(1, 4)):
                s[rng.randrange(len(s))] ^= 1 << rng.randrange(8)
        else:
            s = s[:rng.randrange(1, len(s) + 1)]
        cs.append(bytes(s))
    bad += fuzz_slots.check_corrupted(ctx, cs, 1 << 20, seed * 1000 + r, lambda i, st, want, what: print("MISMATCH corrupted stream", r, i, st, want, what, len(cs[i]), cs[i][:24].hex()))
    left = ctx.last_lean_listed()
    lean_left += left
    lean_done += len(cs) - left
    print("round", r, "done, mismatches so far", bad, "; lean instance decoded", lean_done, "streams, left", lean_left, "to the regular kernel", flush=True)
ctx.close()
sys.exit(1 if bad else 0)
