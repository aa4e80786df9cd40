#!/usr/bin/env python3
"""BASELINE config 5 fixtures: K synthetic 1 MiB multi-meta-block text streams.

Text = seeded order-1 word Markov chain trained on the Canterbury texts among the reference fixtures
(tests/golden/data/{alice29.txt,asyoulik.txt,lcet10.txt,plrabn12.txt}), cut to 1 048 576 B; compressed with the
system libbrotlienc 1.0.9 stream API at quality 9, lgwin 22, BROTLI_OPERATION_FLUSH every 131 072 input bytes (>= 8
meta-blocks, tens of block types and switches).  Writes tests/golden/config5/c5_<seed>.compressed and manifest.json
(sha256 + length of the expected output; the expected bytes are regenerated from the compressed stream by the oracle).
"""
import hashlib
import json
import os
import random
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import brotli_enc  # noqa: E402
import oracle_py  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SIZE = 1 << 20
SRC = ["alice29.txt", "asyoulik.txt", "lcet10.txt", "plrabn12.txt"]


def train():
    words = []
    for f in SRC:
        t = open(os.path.join(ROOT, "tests/golden/data", f), "rb").read().decode("latin-1")
        words += re.findall(r"\S+|\s+", t)
    nxt = {}
    for a, b in zip(words, words[1:]):
        nxt.setdefault(a, []).append(b)
    return words, nxt


def gen(seed, words, nxt):
    rng = random.Random(seed)
    out, n = [], 0
    w = rng.choice(words)
    while n < SIZE:
        out.append(w)
        n += len(w)
        cand = nxt.get(w)
        w = rng.choice(cand) if cand and rng.random() > 0.02 else rng.choice(words)
    return "".join(out).encode("latin-1")[:SIZE]


def main():
    words, nxt = train()
    d = os.path.join(ROOT, "tests/golden/config5")
    os.makedirs(d, exist_ok=True)
    man = {"recipe": "libbrotlienc 1.0.9 stream API, quality 9, lgwin 22, FLUSH every 131072 B; text: seeded order-1 "
                     "word Markov chain over the Canterbury fixtures", "streams": []}
    for seed in range(K):
        text = gen(seed, words, nxt)
        comp = brotli_enc.compress(text, quality=9, lgwin=22, flush_every=131072)
        assert brotli_enc.decompress(comp, len(text)) == text
        st, out, stats = oracle_py.decode(comp, want_stats=True)
        assert st == 0 and out == text
        name = "c5_%d" % seed
        open(os.path.join(d, name + ".compressed"), "wb").write(comp)
        man["streams"].append({"name": name, "in_len": len(comp), "out_len": len(text),
                               "sha256": hashlib.sha256(text).hexdigest(), "stats": stats})
        print(name, len(comp), {k: stats[k] for k in ("meta_blocks", "commands", "literals", "copies", "copy_bytes", "dict_refs", "dict_bytes") if k in stats})
    json.dump(man, open(os.path.join(d, "manifest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
