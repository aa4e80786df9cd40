#!/usr/bin/env python3
"""Re-creates tests/golden/regress_late/r04_wide43_1_90.compressed: stream 90 of round 1 of `tools/wide_fuzz.py 3 43` (libbrotlienc 1.0.9,
quality 11, lgwin 18, a flush every 100 000 bytes, 869 459 bytes of Canterbury text) -- the stream on which the round-4 soak found the
late-resume bug.  Replays the fuzzer's random draws without a GPU; checks the result against the oracle."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import brotli_enc  # noqa: E402
import oracle_py  # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "data")
rng = random.Random(43)
corpus = b"".join(open(os.path.join(G, f), "rb").read() for f in ("lcet10.txt", "plrabn12.txt", "alice29.txt", "asyoulik.txt", "mapsdatazrh"))
for r in range(2):
    datas, streams = [], []
    for it in range(160):
        if it % 4 == 3:
            o = rng.randrange(len(corpus) - 20000)
            data = corpus[o:o + rng.randrange(1, 20000)]
            q = rng.randrange(0, 12)
        else:
            n = rng.randrange(150000, len(corpus) if it % 10 == 0 else 600000)
            o = rng.randrange(len(corpus) - n + 1)
            data = corpus[o:o + n]
            if it % 7 == 0:
                data = b"".join(data[k:k + 3000] if (k // 3000) % 2 else data[k:k + 3000].upper() for k in range(0, len(data), 3000))
            q = rng.choice([9, 10, 11, 11])
        npf = rng.choice([None, None, 0, 1, 2, 3])
        nd = None if npf is None else rng.randrange(0, 16) << npf
        lgwin, mode, fe = rng.randrange(16, 25), rng.randrange(3), rng.choice([0, 0, 0, 100000])
        streams.append(brotli_enc.compress(data, quality=q, lgwin=lgwin, mode=mode, npostfix=npf, ndirect=nd, flush_every=fe))
        datas.append(data)
        if r == 1 and it == 90:
            st, out = oracle_py.decode(streams[-1], cap=len(data) + 64)[:2]
            assert st == 0 and out == data and len(data) == 869459
            d = os.path.join(ROOT, "tests", "golden", "regress_late")
            os.makedirs(d, exist_ok=True)
            open(os.path.join(d, "r04_wide43_1_90.compressed"), "wb").write(streams[-1])
            print("wrote", len(streams[-1]), "bytes; quality", q, "lgwin", lgwin, "flush every", fe)
            sys.exit(0)
    caps = [len(x) + rng.randrange(0, 40) for x in datas]  # (the fuzzer's draws between the rounds)
    for it in range(400):
        s = bytearray(rng.choice(streams))
        if rng.random() < 0.6:
            for _ in range(rng.randrange(1, 4)):
                s[rng.randrange(len(s))] ^= 1 << rng.randrange(8)
        else:
            s = s[:rng.randrange(1, len(s) + 1)]
