#!/usr/bin/env python3
"""One-off soak test on a GPU box: many libbrotlienc-made streams (random data kinds, qualities 0-11, window sizes,
NPOSTFIX/NDIRECT, forced flushes), decoded in ragged batches with unaligned output slots, compared with the original
bytes; plus corrupted variants (bit flips / truncation) compared with the oracle's status.  Usage: big_fuzz.py [rounds] [seed]"""
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
for r in range(rounds):
    datas, streams = [], []
    for it in range(1500):
        kind = rng.randrange(7)
        if kind == 0:
            base = rng.choice(pool)
            n = rng.randrange(1, min(len(base), 120000) if it % 50 == 0 else 30000)
            o = rng.randrange(len(base) - n)
            data = base[o:o + n]
        elif kind == 1:
            data = bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 3000)))
        elif kind == 2:
            unit = bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 300)))
            data = (unit * (1 + 40000 // len(unit)))[:rng.randrange(1, 40000)]
        elif kind == 3:
            base = rng.choice(pool)
            data = b"".join(base[o:o + 150] for o in (rng.randrange(len(base) - 150) for _ in range(rng.randrange(1, 80))))
        elif kind == 4:
            data = bytes(rng.choice(b"ab\n ") for _ in range(rng.randrange(1, 12000)))
        elif kind == 5:
            base = rng.choice(pool)
            o = rng.randrange(len(base) - 5000)
            data = base[o:o + rng.randrange(1, 5000)].upper() + base[o:o + rng.randrange(1, 3000)].title()
        else:  # long fills and long far copies: periodic_fill / bulk copy paths
            unit = bytes(rng.getrandbits(8) for _ in range(rng.choice([1, 2, 3, 5, 16, 43, 64, 100, 257, 1000, 5000])))
            n = rng.randrange(1, 300000 if it % 10 == 0 else 40000)
            head = bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 200)))
            data = head + (unit * (1 + n // len(unit)))[:n] + head[::-1] + (unit[::-1] * 40)[:rng.randrange(0, 20000)]
        npf = rng.choice([None, 0, 1, 2, 3])
        nd = None if npf is None else rng.randrange(0, 16) << npf
        comp = brotli_enc.compress(data, quality=rng.randrange(0, 12), lgwin=rng.randrange(10, 25), mode=rng.randrange(3),
                                   npostfix=npf, ndirect=nd, flush_every=rng.choice([0, 0, 0, 300, 4096, 30000]))
        datas.append(data)
        streams.append(comp)
    caps = [len(x) + rng.randrange(0, 40) for x in datas]  # ragged slots -> every 16-byte skew of the output pointer
    outs, status, out_len = ctx.decode_batch(streams, caps)
    for i, (d, o, st) in enumerate(zip(datas, outs, status)):
        if st != 0 or o != d:
            bad += 1
            print("MISMATCH valid stream", r, i, int(st), len(d), streams[i][:24].hex())
    # corrupted variants: status must equal the oracle's
    cs = []
    for it in range(1500):
        s = bytearray(rng.choice(streams))
        if not s:
            continue
        if rng.random() < 0.5:
            for _ in range(rng.randrange(1, 4)):
                s[rng.randrange(len(s))] ^= 1 << rng.randrange(8)
        else:
            s = s[:rng.randrange(1, len(s) + 1)]
        cs.append(bytes(s))
    bad += fuzz_slots.check_corrupted(ctx, cs, 1 << 21, seed * 1000 + r, lambda i, st, want, what: print("MISMATCH corrupted stream", r, i, st, want, what, cs[i][:24].hex()))
    print("round", r, "done, mismatches so far", bad, flush=True)
ctx.close()
sys.exit(1 if bad else 0)
