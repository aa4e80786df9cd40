#!/usr/bin/env python3
"""One-off soak of the wide-LDS kernel on a GPU box: libbrotlienc streams of LARGE text inputs at high quality (the ones
whose meta-blocks carry more prefix-code tables than the regular kernel's LDS table memory holds), decoded in ragged
batches next to small streams, compared with the original bytes; corrupted variants (bit flips / truncation) compared
with the oracle.  Prints how many streams of each batch the wide kernel took.  Usage: wide_fuzz.py [rounds] [seed] [late]
("late": forced flushes every 20 .. 64 KiB, so that the table need changes from meta-block to meta-block and streams are handed
up under way, with their state -- the late list -- at many positions and alignments)"""
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
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = random.Random(seed)
LATE = len(sys.argv) > 3 and sys.argv[3] == "late"
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"  # "big": every tenth stream ONE piece of 1.2 .. 2.5 MiB of text + an ELF image (80 .. 250
                                                  # literal trees, 10 .. 25 k words of tables: the level-4 instance, DESC_GATHER)
elf = open(sys.executable, "rb").read()
texts = [open(os.path.join(G, f), "rb").read() for f in ("lcet10.txt", "plrabn12.txt", "alice29.txt", "asyoulik.txt", "mapsdatazrh")]
corpus = b"".join(texts)
assert brotli_enc.available()
ctx = brx_knobs.context(0)
bad = 0
for r in range(rounds):
    datas, streams = [], []
    for it in range(160):
        if it % 4 == 3:  # small neighbours: they stay in the regular kernel
            o = rng.randrange(len(corpus) - 20000)
            data = corpus[o:o + rng.randrange(1, 20000)]
            q = rng.randrange(0, 12)
        elif BIG and it % 10 == 1:
            a_, b_ = rng.randrange(200000, 900000), rng.randrange(900000, 1700000)
            o1, o2 = rng.randrange(len(corpus) - a_), rng.randrange(max(1, len(elf) - b_))
            data = corpus[o1:o1 + a_] + elf[o2:o2 + b_] if rng.random() < 0.7 else elf[o2:o2 + b_] + corpus[o1:o1 + a_]
            q = rng.choice([5, 9, 10, 11])
        else:
            n = rng.randrange(150000, len(corpus) if it % 10 == 0 else 600000)
            o = rng.randrange(len(corpus) - n + 1)
            data = corpus[o:o + n]
            if it % 7 == 0:  # two kinds of text in turns: more block types, more trees
                data = b"".join(data[k:k + 3000] if (k // 3000) % 2 else data[k:k + 3000].upper() for k in range(0, len(data), 3000))
            q = rng.choice([9, 10, 11, 11])
        npf = rng.choice([None, None, 0, 1, 2, 3])
        nd = None if npf is None else rng.randrange(0, 16) << npf
        streams.append(brotli_enc.compress(data, quality=q, lgwin=rng.randrange(16, 25), mode=rng.randrange(3), npostfix=npf,
                                           ndirect=nd, flush_every=rng.choice([20000, 30011, 40000, 65536] if LATE else [0, 0, 0, 100000])))
        datas.append(data)
    caps = [len(x) + rng.randrange(0, 40) for x in datas]
    outs, status, out_len = ctx.decode_batch(streams, caps)
    wide = [ctx.last_wide_streams(k) for k in (1, 2, 3)] + ["late %d" % ctx.last_late_streams(), "level 4: %d" % ctx.last_level4()]
    for i, (d, o, st) in enumerate(zip(datas, outs, status)):
        if st != 0 or o != d:
            bad += 1
            print("MISMATCH valid stream", r, i, int(st), len(d), streams[i][:24].hex())
    cs = []
    for it in range(400):
        s = bytearray(rng.choice(streams))
        if rng.random() < 0.6:
            for _ in range(rng.randrange(1, 4)):
                s[rng.randrange(len(s))] ^= 1 << rng.randrange(8)
        else:
            s = s[:rng.randrange(1, len(s) + 1)]
        cs.append(bytes(s))
    CAP = 1 << 22 if BIG else 1 << 21
    bad += fuzz_slots.check_corrupted(ctx, cs, CAP, seed * 1000 + r, lambda i, st, want, what: print("MISMATCH corrupted stream", r, i, st, want, what, cs[i][:24].hex()))
    wide2 = [ctx.last_wide_streams(k) for k in (1, 2, 3)]
    print("round", r, "done: levels 1/2/3 were handed", wide, "of", len(streams), "valid and", wide2, "of", len(cs),
          "corrupted streams; mismatches so far", bad, flush=True)
ctx.close()
sys.exit(1 if bad else 0)
