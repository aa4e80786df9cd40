#!/usr/bin/env python3
"""Fixture-free differential fuzz at scale: every round makes 2048 Brotli streams ON THE GPU (brx_generate_batch) from
edited text / random / periodic / spliced sources with a random meta-block size and block switches on or off, checks that
they decode back (HIP path), then corrupts all of them (bit flips, truncation, appended bytes) and holds the HIP path to
the oracle's status and bytes.  Usage: gen_fuzz.py [rounds] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_py  # noqa: E402
from brotli_rs_amd import brx  # noqa: E402
import brx_knobs  # noqa: E402
import fuzz_slots  # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "data")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
pool = [open(os.path.join(G, f), "rb").read() for f in ("alice29.txt", "lcet10.txt", "plrabn12.txt", "asyoulik.txt")]
ctx = brx_knobs.context(0)
bad = total = 0
for r in range(rounds):
    sources = []
    for k in range(2048):
        kind = rng.randrange(5)
        if kind == 0:
            base = rng.choice(pool)
            o = rng.randrange(len(base) - 60000)
            d = bytearray(base[o:o + rng.randrange(1, 60000 if k % 64 == 0 else 8000)])
            for _ in range(rng.randrange(0, 20)):
                d[rng.randrange(len(d))] = rng.randrange(256)
        elif kind == 1:
            d = rng.randbytes(rng.randrange(0, 3000))
        elif kind == 2:
            unit = rng.randbytes(rng.choice((1, 2, 3, 16, 43, 257, 1000)))
            d = (unit * (1 + 20000 // len(unit)))[:rng.randrange(1, 20000)]
        elif kind == 3:
            base = rng.choice(pool)
            d = b"".join(base[o:o + 90] for o in (rng.randrange(len(base) - 90) for _ in range(rng.randrange(1, 60))))
        else:
            base = rng.choice(pool)
            o = rng.randrange(len(base) - 3000)
            d = base[o:o + rng.randrange(1, 3000)] * rng.randrange(1, 12)
        sources.append(bytes(d))
    mb = rng.choice((200, 1000, 4096, 65536, 1 << 20))
    sw = rng.random() < 0.5
    ad = r % 2 == 1  # every other round: the adaptive generator (codes from statistics, context map, real block switches)
    made = ctx.generate_batch(sources, metablock_bytes=mb, switches=sw, adaptive=ad)
    outs, status, out_len = ctx.decode_batch(made, [len(s) + rng.randrange(0, 40) for s in sources])
    for i, (src, o, st) in enumerate(zip(sources, outs, status)):
        total += 1
        if st != 0 or o != src:
            bad += 1
            print("MISMATCH round trip", r, i, int(st), len(src), mb, sw, ad)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "genfuzz_fail_%d_%d.src" % (r, i)), "wb").write(src)
            open(os.path.join(ROOT, "gpurun_out", "genfuzz_fail_%d_%d.compressed" % (r, i)), "wb").write(made[i])
    cs = []
    for s in made:
        m = bytearray(s)
        x = rng.random()
        if x < 0.6:
            for _ in range(rng.randrange(1, 4)):
                p = rng.randrange(len(m) * 8)
                m[p >> 3] ^= 1 << (p & 7)
        elif x < 0.85:
            m = m[:rng.randrange(1, len(m) + 1)]
        else:
            m += rng.randbytes(rng.randrange(1, 6))
        cs.append(bytes(m))
    total += len(cs)
    bad += fuzz_slots.check_corrupted(ctx, cs, 1 << 19, seed * 1000 + r, lambda i, st, want, what: print("MISMATCH corrupted", r, i, st, want, what, mb, sw, ad, cs[i][:16].hex()))
    print("round %d (meta-blocks of %d, %s): %d streams so far, %d mismatches" % (r, mb, "adaptive" if ad else "switches %s" % sw, total, bad), flush=True)
ctx.close()
sys.exit(1 if bad else 0)
